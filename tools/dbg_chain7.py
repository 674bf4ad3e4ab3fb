"""Debug helper: structure of a mismatch of the 7x7 TAIL chain test (which images / channel tiles / pixels differ from the oracle)."""
import sys
import numpy as np
import torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_chain as tc

def report(got, want, err_msg=''):
    got = np.asarray(got); want = np.asarray(want)
    bad = got != want
    if got.ndim == 4:
        nz = np.nonzero(bad)
        print(f'  bad {bad.sum()} of {bad.size}; per image {bad.reshape(got.shape[0], -1).sum(1)}; channel tiles {np.unique(nz[1] // 32)[:40]}; pixels {np.unique(nz[2] * got.shape[3] + nz[3])[:50]}')
        if bad.any():
            for n in range(min(got.shape[0], 4)):
                print('   image', n, 'bad fraction per pixel (x10):', ' '.join(str(int(10 * v)) if v < 1 else 'X' for v in bad[n].reshape(got.shape[1], -1).mean(0)))
            i = tuple(a[0] for a in nz)
            print('  first bad', i, 'got', got[i], 'want', want[i])
    else:
        print('  bad', bad.sum())
tc.np.testing.assert_array_equal = report
dev = torch.device('cuda:0')
for variant in sys.argv[1:] or ['requant_float=1', 'body_shifts_left']:
    for it in range(2):
        print(variant, 'run', it)
        tc.test_stage_chain_opened_by_a_stride2_block_matches_oracle(dev, (2048, 512, 7, int(__import__('os').environ.get('NID','1')), 1024, 5), variant)

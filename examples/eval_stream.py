"""Streaming evaluation of an F8Net integer model from HOST-resident uint8 batches (the caller side of the reference's test epoch,
/root/reference/fix_train.py:959-969 + forward_loss :676-718): img/s including the host-to-device copies, top-1 / top-5 when labels
are given.

    python examples/eval_stream.py                                   # synthetic ResNet-50 (real fraclen table, random weights), 40 batches of 128
    python examples/eval_stream.py --arch resnet18 --data /path/to/imagenet/val --params model.npz
        --data    ImageFolder-style directory (val/<class>/<image>); needs Pillow; Resize(256) + CenterCrop(224)
        --params  exported IntModel parameters as .npz (keys as the reference's state_dict: f8net_amd.export / onnx_import write them)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from f8net_amd import stream_eval, synth, topology          # noqa: E402
from f8net_amd.net import build_net                            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='resnet50')
    ap.add_argument('--bs', type=int, default=128)
    ap.add_argument('--batches', type=int, default=40)
    ap.add_argument('--data', default=None)
    ap.add_argument('--params', default=None)
    ap.add_argument('--limit', type=int, default=None)
    args = ap.parse_args()
    normalize = args.arch == 'resnet50'            # the tiny-finetuning ymls of the reference (normalize: True); conventional ones feed 0..255
    spec = topology.get(args.arch, normalize=normalize)
    params = dict(np.load(args.params)) if args.params else synth.reference_params(spec, seed=1234)
    net = build_net(spec, params, max_batch=args.bs, hw=224, options={'whole_batch_launches': 1})
    ev = stream_eval.StreamEvaluator(net, normalize=normalize, mean=stream_eval.IMAGENET_MEAN, std=stream_eval.IMAGENET_STD)
    if args.data:
        batches = stream_eval.folder_batches(args.data, args.bs, limit=args.limit)
    else:
        rng = np.random.default_rng(0)
        pool = [rng.integers(0, 256, (args.bs, 224, 224, 3), dtype=np.uint8) for _ in range(4)]
        lab = [rng.integers(0, spec.num_classes, (args.bs,)) for _ in range(4)]
        batches = ((pool[i % 4], lab[i % 4]) for i in range(args.batches))
    r = ev.run(batches)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == '__main__':
    main()

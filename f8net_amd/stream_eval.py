"""Host-fed streaming evaluation: the caller side of the reference's test epoch.

`run_one_epoch(phase='test')` of /root/reference/fix_train.py:959-969 pulls (images, target) batches from a DataLoader and calls
`forward_loss` (:676-718) on each.  Here the batches are HOST-resident uint8 pixels (what a decoder produces): a copy stream moves
batch i+1 to the device while batch i runs, the run takes the pixels as they are (`f8_net_run_u8`: ToTensor / Normalize / the input
quantisation are a 3 x 256 table inside the input kernel), consecutive runs overlap inside the library (`f8_net_set_pipelined(2)`, the
input handed over with `f8_net_set_input_ready`), and the top-k flags (`f8_topk_correct_f32`) are accumulated on the device: one
host synchronisation per epoch, not per batch.

    ev = StreamEvaluator(net, normalize=True, mean=IMAGENET_MEAN, std=IMAGENET_STD)
    stats = ev.run(batches)          # batches: iterable of (uint8 [n,H,W,3] numpy / CPU tensor, int64 [n] labels or None)

Everything computes in libf8net.so; PyTorch supplies pinned host memory, streams and events.
"""
import time

import numpy as np
import torch

from .pipeline import topk_correct

IMAGENET_MEAN = (0.485, 0.456, 0.406)          # fix_train.py:303-304 (transforms.Normalize)
IMAGENET_STD = (0.229, 0.224, 0.225)


class StreamEvaluator:
    def __init__(self, net, normalize=False, mean=None, std=None, topk=(1, 5), device=None, depth=3, nhwc=True):
        """net: a finalized F8Net (f8net_amd.net.build_net).  depth: host / device buffers in rotation (>= 3: one being filled by the
        producer, one in flight over PCIe, one being read by the run two calls back under pipelining mode 2)."""
        self.net, self.normalize, self.mean, self.std, self.topk, self.nhwc = net, normalize, mean, std, tuple(topk), nhwc
        self.dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.depth = max(3, int(depth))
        C, H, W = net.in_shape
        B = net.max_batch
        shape = (B, H, W, C) if nhwc else (B, C, H, W)
        self.h_img = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self.d_img = [torch.empty(shape, dtype=torch.uint8, device=self.dev) for _ in range(self.depth)]
        self.h_tgt = [torch.empty((B,), dtype=torch.int64).pin_memory() for _ in range(self.depth)]
        self.d_tgt = [torch.empty((B,), dtype=torch.int64, device=self.dev) for _ in range(self.depth)]
        self.d_out = [torch.empty((B, net.out_elems), dtype=torch.float32, device=self.dev) for _ in range(self.depth)]
        self.copy_stream = torch.cuda.Stream(self.dev)
        self.copied = [torch.cuda.Event() for _ in range(self.depth)]       # H2D of slot k done
        self.consumed = [torch.cuda.Event() for _ in range(self.depth)]     # the run + scoring that read slot k are done
        self.hits = torch.zeros((len(self.topk),), dtype=torch.float64, device=self.dev)
        net.upload()
        net.set_pipelined(2)

    def run(self, batches, keep_logits=False):
        """Evaluate every batch.  Returns {'images', 'seconds', 'img_per_s', 'top<k>': accuracy or None, 'logits': [...] if keep_logits}."""
        net, dev = self.net, self.dev
        main = torch.cuda.current_stream(dev)
        self.hits.zero_()
        n_img, n_lab, kept, used = 0, 0, [], [False] * self.depth
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i, (img, tgt) in enumerate(batches):
            k = i % self.depth
            n = int(img.shape[0])
            if used[k]:
                self.consumed[k].synchronize()          # the host may overwrite slot k's pinned buffers only after its H2D AND its run are done
            src = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img
            if src.is_pinned():                         # the producer (decoder) wrote straight into page-locked memory: no staging copy; it
                h_src = src                             # must leave the batch alone until its slot comes round again (`depth` batches later)
            else:
                self.h_img[k][:n].copy_(src)            # pageable source: one host copy into the slot's pinned buffer (~10 GB/s on one core)
                h_src = self.h_img[k][:n]
            if tgt is not None:
                self.h_tgt[k][:n].copy_(torch.as_tensor(tgt, dtype=torch.int64))
            with torch.cuda.stream(self.copy_stream):
                self.d_img[k][:n].copy_(h_src, non_blocking=True)
                if tgt is not None:
                    self.d_tgt[k][:n].copy_(self.h_tgt[k][:n], non_blocking=True)
                self.copied[k].record(self.copy_stream)
            out = self.d_out[k][:n]
            net.run_u8(self.d_img[k][:n], normalize=self.normalize, mean=self.mean, std=self.std, nhwc=self.nhwc, out=out,
                       input_ready=self.copied[k])
            if tgt is not None:
                self.hits += topk_correct(out, self.d_tgt[k][:n], self.topk).sum(dim=1).to(torch.float64)
                n_lab += n
            if keep_logits:
                kept.append(out.clone())
            self.consumed[k].record(main)
            used[k] = True
            n_img += n
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        hits = self.hits.cpu().numpy()
        res = {'images': n_img, 'seconds': dt, 'img_per_s': n_img / dt if dt > 0 else 0.0}
        for j, kk in enumerate(self.topk):
            res[f'top{kk}'] = float(hits[j] / n_lab) if n_lab else None
        if keep_logits:
            res['logits'] = [t.cpu().numpy() for t in kept]
        return res


def folder_batches(root, batch, hw=224, resize=256, limit=None):
    """(uint8 [n,hw,hw,3], int64 [n]) batches from an ImageFolder-style directory (root/<class dir>/<image>), the reference's test
    transform: Resize(256) -> CenterCrop(224) (fix_train.py:313-318; ToTensor / Normalize happen inside the net's input kernel).
    Needs Pillow.  Classes are the sorted directory names, as torchvision.datasets.ImageFolder numbers them."""
    import os
    from PIL import Image
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    files = [(os.path.join(root, c, f), ci) for ci, c in enumerate(classes) for f in sorted(os.listdir(os.path.join(root, c)))]
    if limit:
        files = files[:limit]
    for i in range(0, len(files), batch):
        chunk = files[i:i + batch]
        imgs = np.empty((len(chunk), hw, hw, 3), np.uint8)
        for j, (path, _) in enumerate(chunk):
            im = Image.open(path).convert('RGB')
            w, h = im.size
            s = resize / min(w, h)
            # torchvision Resize(int): the SHORT side becomes `resize`, the other int(resize * long / short) (truncation)
            nw, nh = (resize, int(resize * h / w)) if w <= h else (int(resize * w / h), resize)
            im = im.resize((nw, nh), Image.BILINEAR)
            l, t = int(round((nw - hw) / 2.0)), int(round((nh - hw) / 2.0))
            imgs[j] = np.asarray(im.crop((l, t, l + hw, t + hw)), dtype=np.uint8)
            del s
        yield imgs, np.array([c for _, c in chunk], np.int64)

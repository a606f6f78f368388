"""Generates tests/golden/peaked_input.npz: input images whose heat-maps (seeded synthetic hourglass, seed 0) are
PEAKED -- one sharp maximum per joint map with a large margin over every other cell -- like a trained network's.

Random-weight heat-maps of random images are nearly flat, so their arg-max says little about parity (near-ties flip
under 1e-5 rounding differences).  The trained weights are not available offline; instead the INPUT is optimised
(gradient ascent through the CPU oracle, oracle/hourglass_torch.py) so that every joint map of the fixed seeded network
peaks at a planted cell.  The images are stored as uint8 grey values (the pipeline's own input format: a grey frame
replicated to 3 channels, x = u8 / 255), so the fixture is exact and small; the tests recompute the oracle's heat-maps
from it at run time.

    python tests/golden/make_peaked_input.py [n_images] [steps]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hourglass_torch as oh  # noqa: E402


def planted_cells(rng):
    """19 target cells, at least 6 cells apart and 3 cells from the border of the 64 x 128 map."""
    cells = []
    while len(cells) < 19:
        r, c = int(rng.integers(3, 61)), int(rng.integers(3, 125))
        if all(max(abs(r - a), abs(c - b)) >= 6 for a, b in cells):
            cells.append((r, c))
    return cells


def optimise(net, cells, steps, seed, beta=1.0, log=print):
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn((1, 1, 256, 512), generator=g) * 0.5).requires_grad_(True)
    opt = torch.optim.Adam([z], lr=0.1)
    target = torch.tensor([r * 128 + c for r, c in cells])
    for it in range(steps):
        opt.zero_grad()
        x = torch.sigmoid(z).expand(-1, 3, -1, -1)
        hm = net(x)[0].reshape(19, -1)
        loss = F.cross_entropy(beta * hm, target)
        loss.backward()
        opt.step()
        if it % 10 == 0 or it == steps - 1:
            with torch.no_grad():
                top2 = hm.topk(2, dim=1)
                rel = ((top2.values[:, 0] - top2.values[:, 1]) / hm.abs().max()).min().item()
                log(f"  step {it}: loss {loss.item():.4f}, planted cell is the arg-max for {(top2.indices[:, 0] == target).sum().item()}/19 maps, "
                    f"smallest relative margin {rel:.4f}")
    return (torch.sigmoid(z.detach())[0, 0] * 255.0).round().clamp(0, 255).to(torch.uint8).numpy()


def margins(net, u8):
    """Per joint map: (top-1 value - best value outside the top-1 cell) / max |heat-map|."""
    x = torch.from_numpy(u8.astype(np.float32) / 255.0)[:, :, :, None].expand(-1, -1, -1, 3)
    hm = oh.forward_nhwc(net, x).reshape(len(u8), 19, -1)
    top2 = hm.topk(2, dim=-1).values
    return ((top2[..., 0] - top2[..., 1]) / hm.abs().amax(dim=(1, 2), keepdim=True)[..., 0]).numpy(), hm.argmax(dim=-1).numpy()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    torch.set_num_threads(os.cpu_count() or 1)
    net = oh.build(seed=0)
    for p in net.parameters():
        p.requires_grad_(False)
    rng = np.random.default_rng(2024)
    images, planted = [], []
    for i in range(n):
        cells = planted_cells(rng)
        print(f"image {i}: planting {cells}", flush=True)
        images.append(optimise(net, cells, steps, seed=100 + i, log=lambda s: print(s, flush=True)))
        planted.append(cells)
    u8 = np.stack(images)
    m, top = margins(net, u8)
    print("relative margins after uint8 quantisation: min %.3f median %.3f" % (m.min(), np.median(m)))
    hit = top == np.array([[r * 128 + c for r, c in cells] for cells in planted])
    print("planted cell is the arg-max:", hit.sum(), "of", hit.size)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "peaked_input.npz"), images_u8=u8, planted=np.array(planted, dtype=np.int32),
                        oracle_margin=m.astype(np.float32), weights_seed=np.int32(0))


if __name__ == "__main__":
    main()

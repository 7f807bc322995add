#!/usr/bin/env python3
"""GPU counterpart of the reference's test.py: same five (W,H) sizes, same comparison against
PIL.Image.resize, same thresholds, optional backward check and timing table.

    python tools/harness.py [--mode bilinear|bicubic|nearest] [--image tests/golden/kat_test_png.npz]
                            [--size W H] [--backward] [--bench] [--uint8-mode pil|harness] [--out-dir DIR]

Mirrors test.py:15-21 (sizes), :334-379 (PIL compare + asserts), :381-385 (PNG dump, to a writable directory),
:387-398 (backward), :404-416 (benchmark table).  The image defaults to the decoded copy of the reference's
data/test.png kept in tests/golden/ (the reference checkout does not exist on the GPU box).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SIZES = [(320, 196), (460, 220), (120, 96), (1200, 196), (120, 1200)]  # (W, H), test.py:15-21


THRESHOLDS = {"bilinear": (1.0, 1.0 + 1e-5), "bicubic": (1.0, 20.0)}  # (MAE, max abs err) vs PIL: test.py:370-372, :377-379


def evaluate(rgb, size, mode="bilinear", uint8_mode="harness", pil_dn=None):
    """One size of the reference's check (test.py:334-379) on the GPU: resize the HWC uint8 image `rgb` to size = (W, H) and
    compare with PIL.  uint8_mode "harness" does what test.py does around the op (float(), op, clamp for bicubic, byte());
    "pil" runs the Pillow-exact integer path on the uint8 tensor itself.  `pil_dn` = PIL's HWC result (computed here with
    PIL.Image.resize when not given).  Returns {"mae", "max", "variant", "proto" (HWC uint8 numpy), "pil"}."""
    from interpolate_antialiasing_amd import _lib
    from interpolate_antialiasing_amd import extension_interpolate as aa

    fwd = {"bilinear": aa.linear_forward, "nearest": aa.nearest_forward, "bicubic": aa.cubic_forward}[mode]
    if pil_dn is None:
        from PIL import Image

        resample = {"bilinear": Image.BILINEAR, "nearest": Image.BOX, "bicubic": Image.BICUBIC}[mode]
        pil_dn = np.asarray(Image.fromarray(rgb).resize(tuple(size), resample=resample))
    t_img = torch.from_numpy(np.ascontiguousarray(rgb.transpose(2, 0, 1))).cuda()  # uint8 CHW, channels-first like test.py:339
    inv = [size[1], size[0]]
    if uint8_mode == "harness":
        out = fwd(t_img[None].float(), inv, False)
        if mode == "bicubic":
            out = torch.clamp(out, 0, 255)  # test.py:72
        proto = out[0].byte()
    else:
        proto = fwd(t_img[None], inv, False, uint8_mode="pil")[0]
    variant = _lib.last_variant()
    proto = proto.permute(1, 2, 0).cpu().numpy()
    err = np.abs(proto.astype(np.float64) - np.asarray(pil_dn).astype(np.float64))
    return {"mae": float(err.mean()), "max": float(err.max()), "variant": variant, "proto": proto, "pil": np.asarray(pil_dn)}


def main():
    ap = argparse.ArgumentParser("Antialiased interpolation on MI355X vs PIL")
    ap.add_argument("--mode", default="bilinear", choices=["bilinear", "nearest", "bicubic"])
    ap.add_argument("--image", default=os.path.join(ROOT, "tests", "golden", "kat_test_png.npz"))
    ap.add_argument("--size", type=int, nargs=2)
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--bench", action="store_true")
    ap.add_argument("--uint8-mode", default="harness", choices=["pil", "harness"],
                    help="harness = float()/op/byte() exactly as test.py:52-58,75; pil = Pillow-exact integer path")
    ap.add_argument("--out-dir", default=None, help="write the down-sampled PNGs here (test.py:381-385)")
    args = ap.parse_args()

    from PIL import Image

    from interpolate_antialiasing_amd import _lib
    from interpolate_antialiasing_amd import extension_interpolate as aa

    assert torch.cuda.is_available(), "the harness needs a GPU"
    if args.image.endswith(".npz"):
        rgb = np.load(args.image)["rgb"]
    else:
        rgb = np.asarray(Image.open(args.image).convert("RGB")).copy()
    pil_img = Image.fromarray(rgb)
    resample = {"bilinear": Image.BILINEAR, "nearest": Image.BOX, "bicubic": Image.BICUBIC}[args.mode]
    fwd = {"bilinear": aa.linear_forward, "nearest": aa.nearest_forward, "bicubic": aa.cubic_forward}[args.mode]
    t_img = torch.from_numpy(rgb.transpose(2, 0, 1).copy()).cuda()  # uint8 CHW, channels-first like test.py:339
    sizes = [tuple(args.size)] if args.size else SIZES
    rows = []
    for size in sizes:
        inv = [size[1], size[0]]
        r = evaluate(rgb, size, args.mode, args.uint8_mode)
        proto = torch.from_numpy(r["proto"]).permute(2, 0, 1).cuda()
        pil_dn = torch.from_numpy(r["pil"].copy()).permute(2, 0, 1).cuda()
        ref = torch.nn.functional.interpolate(t_img[None].float(), size=inv, mode="nearest" if args.mode == "nearest" else args.mode,
                                              **({} if args.mode == "nearest" else {"align_corners": False}))[0].byte()
        mae_t = (pil_dn.float() - ref.float()).abs().mean().item()
        max_t = (pil_dn.float() - ref.float()).abs().max().item()
        mae, mx = r["mae"], r["max"]
        print(f"size {size}: PyTorch(no AA) vs PIL: MAE {mae_t:.4f} Max {max_t:.0f} | ours[{r['variant']}] vs PIL: MAE {mae:.4f} Max {mx:.0f}")
        if args.mode in THRESHOLDS:
            assert mae < THRESHOLDS[args.mode][0] and mx < THRESHOLDS[args.mode][1]
        if args.out_dir:
            os.makedirs(args.out_dir, exist_ok=True)
            Image.fromarray(proto.permute(1, 2, 0).cpu().numpy()).save(
                os.path.join(args.out_dir, f"aa_interp_{args.mode}_output_{size[0]}_{size[1]}.png"))
        if args.backward and args.mode != "nearest":
            x = t_img[None].double().requires_grad_(True)
            op = getattr(torch.ops.extension_interpolate, "linear_forward" if args.mode == "bilinear" else "cubic_forward")
            y = op(x, inv, False)
            y.sum().backward()
            x2 = t_img[None].double().requires_grad_(True)
            torch.nn.functional.interpolate(x2, size=inv, mode=args.mode, align_corners=False, antialias=True).sum().backward()
            print(f"   grads: ours mean {x.grad.mean().item():.6f}  torch(antialias=True) mean {x2.grad.mean().item():.6f}  "
                  f"max diff {(x.grad - x2.grad).abs().max().item():.2e}")
            assert (x.grad - x2.grad).abs().max().item() < 1e-9
        if args.bench:
            def timed(f, n=50):
                f(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    f()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e6
            xf = t_img[None].float()
            xcl = t_img[None].contiguous(memory_format=torch.channels_last)
            t0 = time.perf_counter(); [pil_img.resize(size, resample=resample) for _ in range(10)]
            rows.append((size, (time.perf_counter() - t0) / 10 * 1e6,
                         timed(lambda: torch.nn.functional.interpolate(xf, size=inv, mode="bilinear", align_corners=False).byte()),
                         timed(lambda: fwd(xf, inv, False)), timed(lambda: fwd(xcl, inv, False, uint8_mode="pil"))))
    if rows:
        print(f"\n{'size':>14} | {'PIL (CPU)':>10} | {'torch no-AA GPU':>15} | {'ours fp32':>10} | {'ours u8 CL':>10}   (us per image, B=1)")
        for size, a, b, c, d in rows:
            print(f"{str(size):>14} | {a:10.1f} | {b:15.1f} | {c:10.1f} | {d:10.1f}")


if __name__ == "__main__":
    main()

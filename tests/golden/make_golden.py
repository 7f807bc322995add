#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD CONTAINER only).

Sources of truth used here, none of which travel to the GPU box:
  * the reference's own C++ compiled by ``make -C oracle ref`` (oracle/_ref/ref_s22.so ...), driven through its
    pybind surface (linear_forward / cubic_forward / nearest_forward / linear_backward);
  * the reference's data files /root/reference/data/test.png and proto_aa_interp_lin_step_one_output.png
    (known-answer test, test.py:324,381-385);
  * Pillow (PIL.Image.resize) — the run-time oracle of the reference's test.py:336;
  * torch.nn.functional.interpolate(antialias=True) autograd in fp64 — the true AA adjoint.

Everything written is data (inputs + expected outputs) in .npz; no reference source text is stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF_DATA = "/root/reference/data"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


HARNESS_SIZES = [(320, 196), (460, 220), (120, 96), (1200, 196), (120, 1200)]  # (W, H): the reference's test.py:15-21


def make_harness():
    """6. What the reference's harness compares against (test.py:336,370-379): PIL.Image.resize of data/test.png at the five
    (W, H) sizes of test.py:15-21, bilinear and bicubic -> harness_pil.npz.  Also checks, right here, that the reference build
    driven the way test.py drives it (float(), op, clamp for bicubic, byte()) meets test.py's thresholds on every size and that
    the oracle's restatements agree with it / with Pillow bit for bit."""
    ref = oracle.load_ref("ref_s22")
    assert ref is not None, "run `make -C oracle ref` first"
    img = np.asarray(Image.open(os.path.join(REF_DATA, "test.png")).convert("RGB")).copy()
    chw = img.transpose(2, 0, 1)[None]
    out = {}
    for (w, h) in HARNESS_SIZES:
        for filt, res, fn in (("linear", Image.BILINEAR, ref.linear_forward), ("cubic", Image.BICUBIC, ref.cubic_forward)):
            pil = np.asarray(Image.fromarray(img).resize((w, h), resample=res))
            out[f"pil_{filt}_{w}x{h}"] = pil
            y = fn(t(chw).float(), [h, w], False)
            if filt == "cubic":
                y = torch.clamp(y, 0, 255)
            proto = y[0].byte().permute(1, 2, 0).numpy()
            err = np.abs(proto.astype(np.float64) - pil.astype(np.float64))
            assert err.mean() < 1.0 and err.max() < (1.0 + 1e-5 if filt == "linear" else 20.0), (filt, w, h, err.mean(), err.max())
            assert np.array_equal(oracle.harness_u8(filt, chw, (h, w))[0].transpose(1, 2, 0), proto), (filt, w, h)
            assert np.array_equal(oracle.pil_resize_u8(filt, chw, (h, w))[0].transpose(1, 2, 0), pil), (filt, w, h)
            out[f"refmae_{filt}_{w}x{h}"] = np.float64(err.mean())
    np.savez_compressed(os.path.join(HERE, "harness_pil.npz"), **out)
    print("harness: PIL outputs of test.png at", len(HARNESS_SIZES), "sizes x {bilinear, bicubic}; reference build within test.py's thresholds")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "harness":
        return make_harness()
    ref = oracle.load_ref("ref_s22")
    ref3 = oracle.load_ref("ref_s3")
    ref3s = oracle.load_ref("ref_s3sep")
    assert ref is not None and ref3 is not None and ref3s is not None, "run `make -C oracle ref` first"
    fwd = {"linear": ref.linear_forward, "cubic": ref.cubic_forward, "box": ref.nearest_forward}
    rng = np.random.default_rng(20211004)

    # ---------------------------------------------------------------- 1. weight tables, pinned by impulse response
    # The reference never exposes its tables, so read them off its output: resizing the identity matrix
    # [1,1,n,n] -> [n,out] leaves H untouched (scale 1 => weights {1,0}, exact) and returns the dense
    # W-pass weight matrix.  The oracle's table must reproduce it bit for bit; the fixture stores the table.
    tables = {}
    combos = [
        ("linear", 906, 320), ("linear", 438, 196), ("linear", 438, 320), ("linear", 906, 196),
        ("cubic", 1024, 224), ("cubic", 906, 320), ("cubic", 438, 196),
        ("linear", 906, 460), ("linear", 438, 220), ("linear", 906, 120), ("linear", 438, 96),
        ("linear", 906, 1200), ("linear", 438, 1200), ("cubic", 906, 1200),
        ("linear", 64, 64), ("linear", 61, 1), ("cubic", 5, 1), ("linear", 3, 2), ("cubic", 3, 2),
        ("box", 906, 320), ("box", 30, 40), ("linear", 53, 23), ("cubic", 53, 23), ("linear", 30, 25),
    ]
    for filt, n, out in combos:
        for dt in (np.float32, np.float64):
            eye = np.eye(n, dtype=dt)[None, None]
            dense = fwd[filt](t(eye), [n, out], False).numpy()[0, 0]  # [n(in x), out]
            k, xmin, xsize, w = oracle.weights(filt, n, out, False, dt)
            mine = np.zeros((n, out), dt)
            for i in range(out):
                for j in range(max(int(xsize[i]), 1)):
                    mine[xmin[i] + j, i] += w[i, j]
            assert np.array_equal(mine, dense), (filt, n, out, dt, np.abs(mine - dense).max())
            key = f"{filt}_{n}_{out}_{np.dtype(dt).name}"
            tables[key + "_ksize"] = np.int64(k)
            tables[key + "_xmin"] = xmin.astype(np.int32)
            tables[key + "_xsize"] = xsize.astype(np.int32)
            tables[key + "_w"] = w
    np.savez_compressed(os.path.join(HERE, "ref_tables.npz"), **tables)
    print("tables:", len(combos), "combos pinned against the reference impulse response")

    # ---------------------------------------------------------------- 2. forward outputs of the reference itself
    out = {}
    cases = [
        ("a", (2, 3, 61, 53), (17, 23), False),     # down/down (SURVEY §8c goldens (2))
        ("b", (1, 2, 20, 30), (40, 25), False),     # H up, W down
        ("c", (1, 3, 33, 47), (33, 90), False),     # H same, W up
        ("d", (2, 1, 12, 17), (5, 7), False),       # gradcheck shape
        ("e", (1, 3, 31, 29), (11, 40), True),      # align_corners=True
        ("f", (1, 4, 64, 64), (3, 200), False),     # extreme
        ("g", (3, 3, 7, 5), (1, 1), False),         # out=1
    ]
    for name, shp, osz, ac in cases:
        x = (rng.random(shp, dtype=np.float32) * 255).astype(np.float32)
        # low-frequency + checkerboard component: random noise alone under-tests window placement
        yy, xx = np.meshgrid(np.arange(shp[2]), np.arange(shp[3]), indexing="ij")
        x += (40 * np.sin(yy / 5.0) * np.cos(xx / 7.0) + 30 * ((yy + xx) % 2)).astype(np.float32)
        out[f"{name}_x"] = x
        out[f"{name}_size"] = np.asarray(osz)
        out[f"{name}_align"] = np.asarray(int(ac))
        for filt in ("linear", "cubic", "box"):
            y32 = fwd[filt](t(x), list(osz), ac).numpy()
            y64 = fwd[filt](t(x.astype(np.float64)), list(osz), ac).numpy()
            ycl = fwd[filt](t(x).contiguous(memory_format=torch.channels_last), list(osz), ac)
            assert np.array_equal(ycl.numpy(), y32)
            assert np.array_equal(oracle.forward(filt, x, osz, ac), y32)
            assert np.array_equal(oracle.forward(filt, x.astype(np.float64), osz, ac), y64)
            out[f"{name}_{filt}_f32"] = y32
            out[f"{name}_{filt}_f64"] = y64
        if not ac and osz[0] < shp[2] and osz[1] < shp[3]:
            # step_three (scale>1 only, bilinear only): both builds agree with step_two_dot_two to 0.0
            assert np.array_equal(ref3.forward(t(x), list(osz), False).numpy(), out[f"{name}_linear_f32"])
            assert np.array_equal(ref3s.forward(t(x), list(osz), False).numpy(), out[f"{name}_linear_f32"])
    np.savez_compressed(os.path.join(HERE, "ref_forward.npz"), **out)
    print("forward:", len(cases), "cases x 3 filters x {f32,f64}")

    # ---------------------------------------------------------------- 3. known-answer test from the reference's data/
    img = np.asarray(Image.open(os.path.join(REF_DATA, "test.png")).convert("RGB")).copy()  # [438,906,3]
    kat1 = np.asarray(Image.open(os.path.join(REF_DATA, "proto_aa_interp_lin_step_one_output.png")))
    kat2 = np.asarray(Image.open(os.path.join(REF_DATA, "proto_aa_interp_lin_step_two_output.png")))
    assert np.array_equal(kat1, kat2) and kat1.shape == (196, 320, 3)
    y = ref.linear_forward(t(img.transpose(2, 0, 1))[None].float(), [196, 320], False)[0].byte().permute(1, 2, 0).numpy()
    assert np.array_equal(y, kat1), "reference build does not reproduce its own committed PNG"
    kat = {"rgb": img, "lin_320x196_u8": kat1}
    kat["lin_320x196_f32"] = ref.linear_forward(t(img.transpose(2, 0, 1))[None].float(), [196, 320], False).numpy()
    kat["cubic_320x196_f32"] = ref.cubic_forward(t(img.transpose(2, 0, 1))[None].float(), [196, 320], False).numpy()
    kat["pil_lin_320x196"] = np.asarray(Image.fromarray(img).resize((320, 196), resample=Image.BILINEAR))
    kat["pil_cubic_320x196"] = np.asarray(Image.fromarray(img).resize((320, 196), resample=Image.BICUBIC))
    np.savez_compressed(os.path.join(HERE, "kat_test_png.npz"), **kat)
    print("KAT: reference PNG reproduced bit for bit by the reference build")

    # ---------------------------------------------------------------- 4. Pillow outputs (uint8 ground truth)
    pil = {"pillow_version": np.asarray(Image.__version__ if hasattr(Image, "__version__") else "unknown")}
    import PIL

    pil["pillow_version"] = np.asarray(PIL.__version__)
    crop = img[170:266, 400:528].copy()  # 96 x 128 crop of test.png
    noise = rng.integers(0, 256, (61, 53, 3), dtype=np.uint8)
    gray = rng.integers(0, 256, (50, 70), dtype=np.uint8)
    pil["crop"], pil["noise"], pil["gray"] = crop, noise, gray
    res = {"linear": Image.BILINEAR, "cubic": Image.BICUBIC, "box": Image.BOX}
    # the five test.py sizes (test.py:15-21, given as (W,H)) scaled to the crop, plus edge cases
    crop_sizes = [(45, 43), (65, 48), (17, 21), (170, 43), (17, 263), (128, 50), (40, 96), (1, 1), (128, 96)]
    for (ow, oh) in crop_sizes:
        for f, r in res.items():
            pil[f"crop_{ow}x{oh}_{f}"] = np.asarray(Image.fromarray(crop).resize((ow, oh), resample=r))
    for (ow, oh) in [(23, 17), (25, 40), (90, 33), (2, 3)]:
        for f, r in res.items():
            pil[f"noise_{ow}x{oh}_{f}"] = np.asarray(Image.fromarray(noise).resize((ow, oh), resample=r))
            pil[f"gray_{ow}x{oh}_{f}"] = np.asarray(Image.fromarray(gray, "L").resize((ow, oh), resample=r))
    np.savez_compressed(os.path.join(HERE, "pil_outputs.npz"), **pil)
    # check the oracle restatement against every one of them right here
    n_ok = 0
    for key, exp in pil.items():
        parts = key.split("_")
        if len(parts) != 3 or "x" not in parts[1]:
            continue
        src = pil[parts[0]]
        ow, oh = map(int, parts[1].split("x"))
        a = src if src.ndim == 3 else src[..., None]
        got = oracle.pil_resize_u8(parts[2], a.transpose(2, 0, 1)[None], (oh, ow))[0].transpose(1, 2, 0)
        got = got if src.ndim == 3 else got[..., 0]
        assert np.array_equal(got, exp), key
        n_ok += 1
    print("Pillow:", n_ok, "outputs, oracle restatement bit-exact on all")

    # ---------------------------------------------------------------- 5. true-adjoint grads (fp64 autograd) + legacy
    bw = {}
    for name, (H, W, oH, oW) in {"a": (12, 17, 5, 7), "b": (61, 53, 17, 23), "c": (20, 30, 40, 25)}.items():
        go = rng.standard_normal((2, 3, oH, oW))
        bw[f"{name}_go"] = go
        bw[f"{name}_in_hw"] = np.asarray([H, W])
        for filt, mode in (("linear", "bilinear"), ("cubic", "bicubic")):
            x = torch.zeros(2, 3, H, W, dtype=torch.float64, requires_grad=True)
            F.interpolate(x, size=(oH, oW), mode=mode, align_corners=False, antialias=True).backward(t(go))
            bw[f"{name}_{filt}_gi"] = x.grad.numpy()
            assert np.abs(oracle.backward(filt, go, (H, W)) - x.grad.numpy()).max() < 1e-12
        # what the reference header's linear_backward returns (NON-AA; label: legacy, do not match)
        leg = ref.linear_backward(t(go.astype(np.float32)), [oH, oW], [2, 3, H, W], False).numpy()
        bw[f"{name}_legacy_nonaa_gi"] = leg
        assert np.array_equal(oracle.legacy_nonaa_linear_backward(go.astype(np.float32), (H, W)), leg)
    np.savez_compressed(os.path.join(HERE, "backward.npz"), **bw)
    print("backward: true-adjoint (fp64 autograd) + legacy non-AA pinned")
    make_harness()


if __name__ == "__main__":
    main()

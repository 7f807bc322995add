"""CPU: the oracle restatement (oracle/aa_oracle.c) against the committed golden vectors.

The fixtures were produced by tests/golden/make_golden.py from the reference's own compiled C++, the
reference's data/ PNGs, Pillow and fp64 autograd; this file re-checks the oracle against them wherever it
runs (build container and GPU box), so the checker itself is pinned before any GPU parity test trusts it.
"""
import os

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FILTS = ("linear", "cubic", "box")


def _table_keys(golden_tables):
    keys = sorted({k.rsplit("_", 1)[0] for k in golden_tables.files})
    return keys


def test_weight_tables_bit_exact(golden_tables):
    n = 0
    for key in _table_keys(golden_tables):
        filt, n_in, n_out, dt = key.split("_")
        k, xmin, xsize, w = oracle.weights(filt, int(n_in), int(n_out), False, np.dtype(dt))
        assert k == int(golden_tables[key + "_ksize"]), key
        assert np.array_equal(xmin, golden_tables[key + "_xmin"]), key
        assert np.array_equal(xsize, golden_tables[key + "_xsize"]), key
        assert np.array_equal(w, golden_tables[key + "_w"]), key
        n += 1
    assert n >= 40


def test_headline_ksize():
    # A -> ksize 7/7, C -> 21
    assert oracle.ksize("linear", 906, 320) == 7
    assert oracle.ksize("linear", 438, 196) == 7
    assert oracle.ksize("cubic", 1024, 224) == 21
    # up-scaling: plain interp_size/2 support (s2.2:208-210)
    assert oracle.ksize("linear", 438, 1200) == 3
    assert oracle.ksize("cubic", 438, 1200) == 5


@pytest.mark.parametrize("case", list("abcdefg"))
def test_forward_bit_exact_vs_reference(golden_forward, case):
    x = golden_forward[f"{case}_x"]
    size = tuple(int(v) for v in golden_forward[f"{case}_size"])
    ac = bool(golden_forward[f"{case}_align"])
    for filt in FILTS:
        assert np.array_equal(oracle.forward(filt, x, size, ac), golden_forward[f"{case}_{filt}_f32"])
        assert np.array_equal(oracle.forward(filt, x.astype(np.float64), size, ac), golden_forward[f"{case}_{filt}_f64"])
        # channels_last input -> channels_last output, same values (SURVEY §4 layout coverage)
        xcl = oracle.as_channels_last(x)
        ycl = oracle.forward(filt, xcl, size, ac)
        assert np.array_equal(ycl, golden_forward[f"{case}_{filt}_f32"])
        if x.shape[1] > 1:
            assert ycl.transpose(0, 2, 3, 1).flags["C_CONTIGUOUS"]
        # threaded port gives the same bits (used by bench.py's cpu_baseline)
        assert np.array_equal(oracle.forward(filt, x, size, ac, nthreads=4), golden_forward[f"{case}_{filt}_f32"])


def test_known_answer_png(golden_kat):
    """The reference's committed data/proto_aa_interp_lin_step_one_output.png (test.py:381-385)."""
    rgb = golden_kat["rgb"]  # [438,906,3]
    x = rgb.transpose(2, 0, 1)[None]
    y = oracle.harness_u8("linear", x, (196, 320))[0].transpose(1, 2, 0)
    assert np.array_equal(y, golden_kat["lin_320x196_u8"])
    yf = oracle.forward("linear", np.ascontiguousarray(x).astype(np.float32), (196, 320))
    assert np.array_equal(yf, golden_kat["lin_320x196_f32"])
    yc = oracle.forward("cubic", np.ascontiguousarray(x).astype(np.float32), (196, 320))
    assert np.array_equal(yc, golden_kat["cubic_320x196_f32"])
    # test.py:370-372 thresholds vs Pillow
    d = np.abs(y.astype(int) - golden_kat["pil_lin_320x196"].astype(int))
    assert d.mean() < 1.0 and d.max() < 1.0 + 1e-5
    # test.py:377-379 (bicubic, clamped then truncated)
    yb = oracle.harness_u8("cubic", x, (196, 320))[0].transpose(1, 2, 0)
    d = np.abs(yb.astype(int) - golden_kat["pil_cubic_320x196"].astype(int))
    assert d.mean() < 1.0 and d.max() < 20.0


def test_pil_semantics_bit_exact(golden_pil, golden_kat):
    n = 0
    for key in golden_pil.files:
        parts = key.split("_")
        if len(parts) != 3 or "x" not in parts[1]:
            continue
        src = golden_pil[parts[0]]
        ow, oh = map(int, parts[1].split("x"))
        a = src if src.ndim == 3 else src[..., None]
        got = oracle.pil_resize_u8(parts[2], a.transpose(2, 0, 1)[None], (oh, ow))[0].transpose(1, 2, 0)
        got = got if src.ndim == 3 else got[..., 0]
        assert np.array_equal(got, golden_pil[key]), key
        n += 1
    assert n >= 50
    x = golden_kat["rgb"].transpose(2, 0, 1)[None]
    for filt, k in (("linear", "pil_lin_320x196"), ("cubic", "pil_cubic_320x196")):
        got = oracle.pil_resize_u8(filt, x, (196, 320), nthreads=4)[0].transpose(1, 2, 0)
        assert np.array_equal(got, golden_kat[k])


def test_pil_layouts_agree():
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (2, 3, 40, 37), dtype=np.uint8)
    a = oracle.pil_resize_u8("linear", x, (13, 19))
    b = oracle.pil_resize_u8("linear", oracle.as_channels_last(x), (13, 19))
    assert np.array_equal(a, b)
    assert b.transpose(0, 2, 3, 1).flags["C_CONTIGUOUS"]


@pytest.mark.parametrize("case", list("abc"))
def test_true_adjoint_vs_autograd(golden_backward, case):
    go = golden_backward[f"{case}_go"]
    hw = tuple(int(v) for v in golden_backward[f"{case}_in_hw"])
    for filt in ("linear", "cubic"):
        gi = oracle.backward(filt, go, hw)
        assert np.abs(gi - golden_backward[f"{case}_{filt}_gi"]).max() < 1e-12
        gi32 = oracle.backward(filt, go.astype(np.float32), hw)
        assert np.abs(gi32 - golden_backward[f"{case}_{filt}_gi"]).max() < 1e-4
    # the header's backward is NOT the AA adjoint: reproduced only to pin that statement
    leg = oracle.legacy_nonaa_linear_backward(go.astype(np.float32), hw)
    assert np.array_equal(leg, golden_backward[f"{case}_legacy_nonaa_gi"])
    assert np.abs(leg - golden_backward[f"{case}_linear_gi"]).max() > 0.1


def test_adjoint_identity():
    rng = np.random.default_rng(3)
    for filt in ("linear", "cubic", "box"):
        x = rng.standard_normal((2, 2, 23, 31))
        g = rng.standard_normal((2, 2, 9, 40))
        lhs = (oracle.forward(filt, x, (9, 40)) * g).sum()
        rhs = (x * oracle.backward(filt, g, (23, 31))).sum()
        assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_empty_batch_and_errors():
    x = np.zeros((0, 3, 8, 8), np.float32)
    assert oracle.forward("linear", x, (4, 4)).shape == (0, 3, 4, 4)
    with pytest.raises(RuntimeError):
        oracle.forward("linear", np.zeros((1, 1, 4, 4), np.float32), (0, 4))


def test_oracle_edge_cases_under_asan():
    """The reference found its one memory bug with AddressSanitizer (README.md:507-520: step_two_dot_one's `j < 2` unroll read a tap
    past a one-tap window at the last row; fixed at step_two_dot_two/aa_interpolation_impl.h:45-51,75-80).  The C restatement gets the
    same treatment: `make -C oracle asan` builds it with -fsanitize=address,undefined next to a driver (oracle/asan_edge.c) that runs
    out = 1, in < ksize, in = out, up-scaling and one-pixel-wide shapes x 3 filters x align_corners through every entry point with
    exact-size heap buffers; any access one element past a window aborts the run."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    odir = os.path.join(ROOT, "oracle")
    b = subprocess.run(["make", "-C", odir, "asan"], capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in (b.stderr + b.stdout) and "cannot find" in (b.stderr + b.stdout):
        pytest.skip("this toolchain has no libasan")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([os.path.join(odir, "_asan", "asan_edge")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="2"))
    assert r.returncode == 0 and "ASAN_EDGE_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])

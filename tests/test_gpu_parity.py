"""GPU parity tests (-m gpu): the HIP path, called through the Python shim -> C-ABI, against the oracle,
the committed golden vectors and size-independent properties at BASELINE.json's full sizes.

Bars: bit-exact for uint8 (Pillow semantics and harness semantics) and for the weight tables; fp32/fp64
forward is ALSO held bit-exact against the reference build's outputs (the kernels round product and sum
separately, in tap order, like the reference's CPU code) — the north-star tolerance is 1e-4 relative.
"""
import os
import sys

import numpy as np
import pytest
import torch

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

FILTS = ("linear", "cubic", "box")


@pytest.fixture(scope="module")
def aa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from interpolate_antialiasing_amd import extension_interpolate

    return extension_interpolate


def _fn(aa, filt):
    return {"linear": aa.linear_forward, "cubic": aa.cubic_forward, "box": aa.nearest_forward}[filt]


def _gpu(a, channels_last=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if channels_last:
        t = t.contiguous(memory_format=torch.channels_last)
    return t


# ------------------------------------------------------------------------------------------------ tables
def test_device_tables_bit_exact(aa, golden_tables):
    from interpolate_antialiasing_amd import _lib, tables

    n = 0
    for key in sorted({k.rsplit("_", 1)[0] for k in golden_tables.files}):
        filt, n_in, n_out, dt = key.split("_")
        kind = _lib.TABLE_F32 if dt == "float32" else _lib.TABLE_F64
        t = tables.build_table(oracle.FILTERS[filt], kind, int(n_in), int(n_out), False, 0.0, torch.device("cuda"))
        xmin, xsize, w = t.unpack()
        assert t.ksize == int(golden_tables[key + "_ksize"]), key
        assert np.array_equal(xmin, golden_tables[key + "_xmin"]), key
        assert np.array_equal(xsize, golden_tables[key + "_xsize"]), key
        assert np.array_equal(w, golden_tables[key + "_w"]), key
        assert t.max_taps == max(1, int(xsize.max())), key
        sc = t.unpack_scatter()  # float tables carry the adjoint-form records too: 32-byte (float) or 64-byte (double) records
        if sc is not None and t.scatter_max <= 6:
            first, count, sw, completes = sc
            assert sw.dtype == (np.float32 if kind == _lib.TABLE_F32 else np.float64)
            last = xmin + np.maximum(xsize, 1) - 1
            dense = np.zeros((int(n_in), int(n_out)), w.dtype)
            for o in range(int(n_out)):
                dense[xmin[o]:xmin[o] + xsize[o], o] = w[o, :xsize[o]]
            for x in range(int(n_in)):
                fed = np.nonzero((xmin <= x) & (last >= x))[0]
                assert count[x] == len(fed) and (len(fed) == 0 or first[x] == fed[0]), (key, x)
                assert np.array_equal(sw[x, :count[x]], dense[x, first[x]:first[x] + count[x]]) and not sw[x, count[x]:].any(), (key, x)
            assert completes.sum() == int(n_out), key
        n += 1
    assert n >= 40


@pytest.mark.parametrize("filt", FILTS)
def test_device_pil_tables_bit_exact(aa, filt):
    from interpolate_antialiasing_amd import _lib, tables

    for n_in, n_out in [(906, 320), (438, 196), (1024, 224), (906, 1200), (438, 1200), (53, 23), (61, 1), (3, 2), (64, 64)]:
        k, xmin, xsize, kk, _ = oracle.pil_coeffs(filt, n_in, n_out)
        t = tables.build_table(oracle.FILTERS[filt], _lib.TABLE_PIL, n_in, n_out, False, 0.0, torch.device("cuda"))
        dxmin, dxsize, dw = t.unpack()
        assert t.ksize == k
        assert np.array_equal(dxmin, xmin) and np.array_equal(dxsize, xsize) and np.array_equal(dw, kk), (filt, n_in, n_out)
        sc = t.unpack_scatter()
        if sc is not None and t.scatter_max <= t.scatter_ksize:  # the adjoint-form records the fused kernels read
            first, count, w, completes = sc
            last = xmin + np.maximum(xsize, 1) - 1  # last input index of every output's window
            dense = np.zeros((n_in, n_out), np.int64)
            for o in range(n_out):
                dense[xmin[o]:xmin[o] + xsize[o], o] = kk[o, :xsize[o]]
            for x in range(n_in):
                fed = np.nonzero((xmin <= x) & (last >= x))[0]
                assert count[x] == len(fed) and (len(fed) == 0 or first[x] == fed[0]), (filt, n_in, n_out, x)
                assert np.array_equal(w[x, :count[x]], dense[x, first[x]:first[x] + count[x]]) and not w[x, count[x]:].any()
                assert completes[x] == np.count_nonzero(last == x) and (completes[x] == 0 or last[first[x]] == x)
            assert completes.sum() == n_out


def test_align_corners_tables(aa):
    from interpolate_antialiasing_amd import _lib, tables

    for filt in ("linear", "cubic"):
        for dt, kind in ((np.float32, _lib.TABLE_F32), (np.float64, _lib.TABLE_F64)):
            k, xmin, xsize, w = oracle.weights(filt, 31, 11, True, dt)
            t = tables.build_table(oracle.FILTERS[filt], kind, 31, 11, True, 0.0, torch.device("cuda"))
            dxmin, dxsize, dw = t.unpack()
            assert t.ksize == k and np.array_equal(dxmin, xmin) and np.array_equal(dxsize, xsize) and np.array_equal(dw, w)


# ------------------------------------------------------------------------------------------------ forward, float
@pytest.mark.parametrize("case", list("abcdefg"))
@pytest.mark.parametrize("channels_last", [False, True])
def test_forward_vs_reference_golden(aa, golden_forward, case, channels_last):
    x = golden_forward[f"{case}_x"]
    size = [int(v) for v in golden_forward[f"{case}_size"]]
    ac = bool(golden_forward[f"{case}_align"])
    for filt in FILTS:
        for dt in ("f32", "f64"):
            xin = x if dt == "f32" else x.astype(np.float64)
            y = _fn(aa, filt)(_gpu(xin, channels_last), size, ac)
            exp = golden_forward[f"{case}_{filt}_{dt}"]
            assert y.shape == exp.shape
            if channels_last and x.shape[1] > 1:
                assert y.is_contiguous(memory_format=torch.channels_last)  # s2.2:752
            got = y.cpu().numpy()
            # north-star tolerance ...
            np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-4)
            # ... and the stronger bar this build holds: identical bits
            assert np.array_equal(got, exp), (case, filt, dt, float(np.abs(got - exp).max()))


def test_forward_known_answer_and_headline_fp32(aa, golden_kat):
    rgb = golden_kat["rgb"]
    x = np.ascontiguousarray(rgb.transpose(2, 0, 1)[None]).astype(np.float32)  # config 0: fp32 NCHW [1,3,438,906]
    y = aa.linear_forward(_gpu(x), [196, 320], False).cpu().numpy()
    assert np.array_equal(y, golden_kat["lin_320x196_f32"])
    yc = aa.cubic_forward(_gpu(x), [196, 320], False).cpu().numpy()
    assert np.array_equal(yc, golden_kat["cubic_320x196_f32"])
    # the reference's committed PNG = truncation of the fp32 result (test.py:75)
    assert np.array_equal(y[0].astype(np.uint8).transpose(1, 2, 0), golden_kat["lin_320x196_u8"])


@pytest.mark.parametrize("shape,size", [((3, 3, 97, 131), (41, 37)), ((2, 1, 64, 300), (100, 64)), ((1, 5, 9, 1000), (9, 77)),
                                        ((4, 3, 438, 906), (196, 320)), ((1, 3, 1024, 1024), (224, 224))])
def test_forward_random_vs_oracle(aa, shape, size):
    rng = np.random.default_rng(hash((shape, size)) % (2**32))
    x = (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
    for filt in FILTS:
        for cl in (False, True):
            exp = oracle.forward(filt, x, size, nthreads=8)
            got = _fn(aa, filt)(_gpu(x, cl), list(size), False).cpu().numpy()
            assert np.array_equal(got, exp), (filt, cl, float(np.abs(got - exp).max()))


# ------------------------------------------------------------------------------------------------ forward, uint8
def test_u8_pil_golden(aa, golden_pil, golden_kat):
    n = 0
    for key in golden_pil.files:
        parts = key.split("_")
        if len(parts) != 3 or "x" not in parts[1]:
            continue
        src = golden_pil[parts[0]]
        ow, oh = map(int, parts[1].split("x"))
        a = src if src.ndim == 3 else src[..., None]
        # NHWC storage viewed as NCHW = torch channels_last (config 1's layout)
        x = torch.from_numpy(a.copy()).cuda().permute(2, 0, 1)[None]
        y = _fn(aa, parts[2])(x, [oh, ow], False, uint8_mode="pil")
        got = y[0].permute(1, 2, 0).cpu().numpy()
        got = got if src.ndim == 3 else got[..., 0]
        assert np.array_equal(got, golden_pil[key]), key
        # plain NCHW-contiguous input gives the same values
        y2 = _fn(aa, parts[2])(x.contiguous(), [oh, ow], False, uint8_mode="pil")
        assert torch.equal(y2, y.contiguous()), key
        n += 1
    assert n >= 50


def test_u8_headline_config_pil_parity(aa, golden_kat):
    """BASELINE config 1: uint8 channels_last [1,3,438,906] -> [196,320] bilinear, PIL parity (MaxAbsE <= 1.0)."""
    rgb = golden_kat["rgb"]
    x = torch.from_numpy(rgb.copy()).cuda().permute(2, 0, 1)[None]
    assert x.is_contiguous(memory_format=torch.channels_last)
    y = aa.linear_forward(x, [196, 320], False)
    assert y.is_contiguous(memory_format=torch.channels_last)
    got = y[0].permute(1, 2, 0).cpu().numpy()
    d = np.abs(got.astype(int) - golden_kat["pil_lin_320x196"].astype(int))
    assert d.max() <= 1.0
    assert d.max() == 0  # Pillow semantics: exact
    yb = aa.cubic_forward(x, [196, 320], False)[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(yb, golden_kat["pil_cubic_320x196"])
    # harness semantics reproduce the reference's committed PNG bit for bit, and test.py:370-372's thresholds
    yh = aa.linear_forward(x, [196, 320], False, uint8_mode="harness")[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(yh, golden_kat["lin_320x196_u8"])
    d = np.abs(yh.astype(int) - golden_kat["pil_lin_320x196"].astype(int))
    assert d.mean() < 1.0 and d.max() < 1.0 + 1e-5


@pytest.mark.parametrize("shape,size", [((5, 3, 438, 906), (196, 320)), ((3, 3, 906, 438), (320, 196)), ((2, 4, 50, 70), (31, 22)),
                                        ((2, 1, 33, 47), (66, 20)), ((1, 3, 1024, 1024), (224, 224)), ((2, 3, 100, 37), (100, 90)),
                                        ((7, 3, 64, 64), (1, 1)), ((1, 2, 5, 4), (40, 33)),
                                        # byte counts that are not multiples of 4 / 16 on the fused paths (range-checked tails)
                                        ((1, 3, 33, 37), (15, 16)), ((3, 3, 35, 41), (17, 20)), ((2, 4, 31, 29), (9, 11))])
def test_u8_random_vs_oracle(aa, shape, size):
    rng = np.random.default_rng(hash((shape, size)) % (2**32))
    x = rng.integers(0, 256, shape, dtype=np.uint8)
    for filt in FILTS:
        for cl in (False, True):
            xin = oracle.as_channels_last(x) if cl else x
            exp = oracle.pil_resize_u8(filt, xin, size, nthreads=8)
            got = _fn(aa, filt)(_gpu(x, cl), list(size), False, uint8_mode="pil").cpu().numpy()
            assert np.array_equal(got, exp), (filt, cl)
            exph = oracle.harness_u8(filt, xin, size, nthreads=8)
            goth = _fn(aa, filt)(_gpu(x, cl), list(size), False, uint8_mode="harness").cpu().numpy()
            assert np.array_equal(goth, exph), (filt, cl, "harness")


def test_u8_extremes(aa):
    """Saturation paths: all-0, all-255 and checkerboards through the bicubic (negative lobes -> clip8)."""
    for val in (0, 255):
        x = torch.full((2, 3, 64, 80), val, dtype=torch.uint8, device="cuda").contiguous(memory_format=torch.channels_last)
        for f in (aa.linear_forward, aa.cubic_forward, aa.nearest_forward):
            assert int(f(x, [23, 31]).float().sub(val).abs().max()) == 0
    yy, xx = np.meshgrid(np.arange(64), np.arange(80), indexing="ij")
    cb = (((yy // 3 + xx // 3) % 2) * 255).astype(np.uint8)[None, None].repeat(3, 1)
    exp = oracle.pil_resize_u8("cubic", cb, (23, 31))
    got = aa.cubic_forward(_gpu(cb), [23, 31]).cpu().numpy()
    assert np.array_equal(got, exp)


# ------------------------------------------------------------------------------------------------ full-size properties
def test_fullsize_properties(aa):
    """BASELINE sizes, no oracle needed: batch independence, constants preserved (weights sum to 1), linearity."""
    torch.manual_seed(0)
    x = torch.randint(0, 256, (64, 3, 438, 906), dtype=torch.uint8, device="cuda").contiguous(memory_format=torch.channels_last)
    y = aa.linear_forward(x, [196, 320])
    for i in (0, 17, 63):
        assert torch.equal(aa.linear_forward(x[i:i + 1], [196, 320]), y[i:i + 1])
    # a checksum of checksums that must not depend on batch order
    perm = torch.randperm(64, device="cuda")
    yp = aa.linear_forward(x[perm].contiguous(memory_format=torch.channels_last), [196, 320])
    assert torch.equal(yp, y[perm])
    c = torch.full((8, 3, 906, 438), 37, dtype=torch.uint8, device="cuda")
    assert torch.equal(aa.linear_forward(c, [320, 196]), torch.full((8, 3, 320, 196), 37, dtype=torch.uint8, device="cuda"))
    # fp32: linearity F(a x + b z) = a F(x) + b F(z) within rounding, bicubic [*,3,1024,1024] -> [224,224] (config 2)
    a = torch.rand(4, 3, 1024, 1024, device="cuda")
    b = torch.rand(4, 3, 1024, 1024, device="cuda")
    lhs = aa.cubic_forward(2.0 * a - 3.0 * b, [224, 224])
    rhs = 2.0 * aa.cubic_forward(a, [224, 224]) - 3.0 * aa.cubic_forward(b, [224, 224])
    assert float((lhs - rhs).abs().max()) < 1e-4
    ones = torch.ones(2, 3, 1024, 1024, device="cuda")
    assert float((aa.cubic_forward(ones, [224, 224]) - 1).abs().max()) < 1e-5


def test_empty_batch_and_errors(aa):
    x = torch.zeros(0, 3, 8, 8, device="cuda")
    assert tuple(aa.linear_forward(x, [4, 4]).shape) == (0, 3, 4, 4)  # s2.2:747-750
    with pytest.raises(RuntimeError, match="Input and output sizes should be greater than 0"):
        aa.linear_forward(torch.zeros(1, 3, 8, 8, device="cuda"), [0, 4])
    with pytest.raises(RuntimeError, match="It is expected input_size equals to 4"):
        aa.linear_forward(torch.zeros(3, 8, 8, device="cuda"), [4, 4])
    with pytest.raises(RuntimeError, match="It is expected output_size equals to 2"):
        aa.linear_forward(torch.zeros(1, 3, 8, 8, device="cuda"), [4])
    with pytest.raises(NotImplementedError, match="not implemented for 'Int'"):
        aa.linear_forward(torch.zeros(1, 3, 8, 8, device="cuda", dtype=torch.int32), [4, 4])


def test_noncontiguous_input(aa):
    x = torch.rand(2, 3, 40, 60, device="cuda")[:, :, ::2, 5:50]
    exp = oracle.forward("linear", x.cpu().numpy().copy(), (7, 11))
    assert np.array_equal(aa.linear_forward(x, [7, 11]).cpu().numpy(), exp)


# ------------------------------------------------------------------------------------------------ backward
@pytest.mark.parametrize("case", list("abc"))
@pytest.mark.parametrize("atomic", [False, True])
def test_backward_true_adjoint_golden(aa, golden_backward, case, atomic):
    go = golden_backward[f"{case}_go"]
    h, w = (int(v) for v in golden_backward[f"{case}_in_hw"])
    n, c, oh, ow = go.shape
    for filt, bw in (("linear", aa.linear_backward), ("cubic", aa.cubic_backward)):
        exp = golden_backward[f"{case}_{filt}_gi"]  # fp64 autograd of F.interpolate(antialias=True)
        g64 = bw(_gpu(go), [oh, ow], [n, c, h, w], False, atomic=atomic).cpu().numpy()
        assert np.abs(g64 - exp).max() < 1e-11
        g32 = bw(_gpu(go.astype(np.float32)), [oh, ow], [n, c, h, w], False, atomic=atomic).cpu().numpy()
        np.testing.assert_allclose(g32, exp, rtol=1e-4, atol=1e-4)
        gcl = bw(_gpu(go, True), [oh, ow], [n, c, h, w], False, atomic=atomic)
        assert np.abs(gcl.cpu().numpy() - exp).max() < 1e-11
    # and NOT the header's non-AA backward
    assert np.abs(g32 - golden_backward[f"{case}_legacy_nonaa_gi"]).max() > 0.1


def test_backward_fullsize_adjoint_identity(aa):
    """Config 5: grad [1,3,196,320] -> [1,3,438,906]; <F x, g> == <x, F^T g> (size-independent property)."""
    torch.manual_seed(1)
    for fwd, bwd in ((aa.linear_forward, aa.linear_backward), (aa.cubic_forward, aa.cubic_backward)):
        x = torch.randn(1, 3, 438, 906, device="cuda", dtype=torch.float64)
        g = torch.randn(1, 3, 196, 320, device="cuda", dtype=torch.float64)
        lhs = (fwd(x, [196, 320]) * g).sum().item()
        for atomic in (False, True):
            rhs = (x * bwd(g, [196, 320], [1, 3, 438, 906], False, atomic=atomic)).sum().item()
            assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))
        # fp32 against the oracle's adjoint
        g32 = g.float()
        exp = oracle.backward("linear" if fwd is aa.linear_forward else "cubic", g32.cpu().numpy(), (438, 906))
        got = bwd(g32, [196, 320], [1, 3, 438, 906]).cpu().numpy()
        np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-5)


def test_gradcheck_fp64(aa):
    """test.py:394-398's gradcheck (eps=1e-8 there is below fp64 resolution for this op; 1e-6 is used) on the
    [1,2,12,17] -> [5,7] shape of SURVEY §8d(5), through torch.ops + registered autograd."""
    from torch.autograd import gradcheck

    for op in (torch.ops.extension_interpolate.linear_forward, torch.ops.extension_interpolate.cubic_forward,
               torch.ops.extension_interpolate.nearest_forward):
        x = torch.rand(1, 2, 12, 17, device="cuda", dtype=torch.float64, requires_grad=True)
        assert gradcheck(lambda t: op(t, [5, 7], False), (x,), eps=1e-6, atol=1e-6, rtol=1e-6, check_batched_grad=False)
        xu = torch.rand(1, 2, 6, 5, device="cuda", dtype=torch.float64, requires_grad=True)
        assert gradcheck(lambda t: op(t, [9, 11], False), (xu,), eps=1e-6, atol=1e-6, rtol=1e-6, check_batched_grad=False)


def test_backward_shape_errors(aa):
    g = torch.zeros(1, 3, 5, 7, device="cuda")
    with pytest.raises(RuntimeError, match="Expected grad_output to have the same shape as output"):
        aa.linear_backward(g, [5, 8], [1, 3, 12, 17])
    with pytest.raises(RuntimeError, match="dimension 4"):
        aa.linear_backward(g[0], [5, 7], [1, 3, 12, 17])


def test_native_library_is_what_ran(aa):
    """The HIP extension, not a fallback: the .so is mapped into this process and reports its variant."""
    from interpolate_antialiasing_amd import _lib

    x = torch.rand(1, 3, 20, 20, device="cuda")
    aa.linear_forward(x, [7, 9])
    assert _lib.last_variant() != "none"
    with open("/proc/self/maps") as f:
        assert "libaa_interp.so" in f.read()


# ------------------------------------------------------------------------------------------------ fused vs generic
def test_fused_kernels_match_generic_at_full_size(aa):
    """Two independent implementations (fused single-launch vs generic two-launch) must agree bit for bit at
    BASELINE sizes, where the CPU oracle is too slow to be the checker for whole batches."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(3)
    cases = []
    x8 = torch.randint(0, 256, (48, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    cases += [(aa.linear_forward, x8, [196, 320]), (aa.cubic_forward, x8, [196, 320]), (aa.nearest_forward, x8, [196, 320])]
    x8t = torch.randint(0, 256, (16, 906, 438, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)  # config 3 shape
    cases += [(aa.linear_forward, x8t, [320, 196])]
    x4 = torch.randint(0, 256, (5, 300, 500, 4), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)   # C = 4
    cases += [(aa.linear_forward, x4, [111, 204]), (aa.cubic_forward, x4, [150, 252])]
    x8p = torch.randint(0, 256, (12, 3, 438, 906), dtype=torch.uint8, device="cuda")                     # planar (NCHW) bytes
    cases += [(aa.linear_forward, x8p, [196, 320]), (aa.cubic_forward, x8p, [196, 320]), (aa.nearest_forward, x8p[:2], [100, 204]),
              (aa.linear_forward, x8p[:, :1], [438, 320]), (aa.linear_forward, x8p[:3], [196, 322])]
    harness = lambda fn: (lambda x, size: fn(x, size, uint8_mode="harness"))                             # reference harness semantics
    cases += [(harness(aa.linear_forward), x8, [196, 320]), (harness(aa.cubic_forward), x8[:8], [196, 320]),
              (harness(aa.nearest_forward), x8[:8], [196, 320]), (harness(aa.linear_forward), x8t[:4], [320, 196]),
              (harness(aa.cubic_forward), x4, [150, 252]), (harness(aa.linear_forward), x8[:2], [438, 320]),
              (harness(aa.linear_forward), x8p[:4], [196, 320]), (harness(aa.cubic_forward), x8p[:2], [196, 320])]  # test.py's layout
    # rows that are not whole dwords (oW % 4 != 0): byte stores, still one launch
    cases += [(aa.linear_forward, x8[:3], [196, 322]), (aa.cubic_forward, x8[:3], [200, 402]), (harness(aa.linear_forward), x8[:3], [196, 323]),
              (aa.linear_forward, x4[:2], [111, 203]), (aa.linear_forward, x8p[:2], [196, 321])]
    xw = torch.randint(0, 256, (2, 120, 1700, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)  # wide: 10 strips, short last group
    cases += [(aa.linear_forward, xw, [60, 600]), (harness(aa.linear_forward), xw, [60, 600]), (aa.linear_forward, xw.contiguous(), [60, 600])]
    xf = torch.rand(6, 3, 438, 906, device="cuda") * 255
    cases += [(aa.linear_forward, xf, [196, 320]), (aa.cubic_forward, xf, [196, 320]), (aa.nearest_forward, xf, [196, 320])]
    xfl = xf.contiguous(memory_format=torch.channels_last)                                              # fp32 channels_last (s2.2:752)
    cases += [(aa.linear_forward, xfl, [196, 320]), (aa.cubic_forward, xfl, [196, 320]), (aa.nearest_forward, xfl[:2], [200, 300])]
    xf4 = (torch.rand(2, 4, 300, 500, device="cuda") - 0.3).contiguous(memory_format=torch.channels_last)
    cases += [(aa.linear_forward, xf4, [111, 204]), (aa.cubic_forward, xf4, [150, 251])]
    xd = (torch.rand(2, 3, 438, 906, device="cuda", dtype=torch.float64) * 255)                          # fp64 (the reference dispatches double, s2.2:609-614)
    cases += [(aa.linear_forward, xd, [196, 320]), (aa.cubic_forward, xd, [196, 320]), (aa.nearest_forward, xd, [200, 300])]
    xc = torch.rand(3, 3, 1024, 1024, device="cuda") * 255                                              # config 2 shape
    cases += [(aa.cubic_forward, xc, [224, 224]), (aa.linear_forward, xc, [224, 224])]
    xo = torch.rand(2, 2, 333, 517, device="cuda") - 0.5                                                # odd sizes, signed data
    cases += [(aa.cubic_forward, xo, [100, 129]), (aa.linear_forward, xo, [333, 100]), (aa.linear_forward, xo, [77, 517])]
    xu = torch.rand(2, 3, 438, 906, device="cuda") * 255                                                # test.py's up-scales
    cases += [(aa.linear_forward, xu, [1200, 1200]), (aa.cubic_forward, xu, [1200, 1200]), (aa.linear_forward, xo, [400, 300]),
              (aa.nearest_forward, xo, [500, 600]), (aa.cubic_forward, xo, [333, 900])]
    go = torch.randn(8, 3, 196, 320, device="cuda")                                                     # config 5, batched
    bwd = lambda fn, ishape: (lambda g, size: fn(g, size, ishape))
    cases += [(bwd(aa.linear_backward, [8, 3, 438, 906]), go, [196, 320]), (bwd(aa.cubic_backward, [8, 3, 438, 906]), go, [196, 320]),
              (bwd(aa.linear_backward, [8, 3, 196, 906]), go, [196, 320]), (bwd(aa.linear_backward, [8, 3, 500, 320]), go, [196, 320])]
    fused_seen = set()
    try:
        for fn, x, size in cases:
            _lib.set_fused(1)
            y1 = fn(x, size)
            fused_seen.add(_lib.last_variant())
            assert _lib.last_variant().startswith("fused"), (_lib.last_variant(), getattr(fn, "__name__", "?"), tuple(x.shape), size)
            _lib.set_fused(0)
            y0 = fn(x, size)
            assert _lib.last_variant().startswith("generic"), _lib.last_variant()
            assert torch.equal(y1, y0), (getattr(fn, "__name__", "backward"), tuple(x.shape), size)
    finally:
        _lib.set_fused(1)
    assert {"fused_u8_nhwc_pil_v3", "fused_u8_planar_pil_v3", "fused_u8_nhwc_harness_v3", "fused_u8_planar_harness_v3", "fused_f32_nchw",
            "fused_f32_nchw_up", "fused_f32_nhwc", "fused_f64_nchw"} <= fused_seen, fused_seen


def test_all_fused_generations_agree(aa):
    """The first-generation uint8 kernel stays selectable (A/B runs, shapes v3 does not take); keep it correct,
    including the last pixel of the last image (a partially out-of-range dword must not be zeroed)."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(4)
    x = torch.randint(0, 256, (7, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    try:
        _lib.set_fused(0)
        ref = aa.linear_forward(x, [196, 320])
        seen = []
        for mode in (1, 2):
            _lib.set_fused(mode)
            y = aa.linear_forward(x, [196, 320])
            seen.append(_lib.last_variant())
            assert torch.equal(y, ref), (mode, _lib.last_variant())
    finally:
        _lib.set_fused(1)
    assert seen == ["fused_u8_nhwc_pil_v3", "fused_u8_nhwc_pil"], seen


def test_dispatch_does_not_depend_on_pointers(aa):
    """aa_workspace_bytes() answers from the shape alone, so no kernel may decline on a pointer: a uint8 output that starts on an odd byte
    (a sliced view handed to the C-ABI) is served by the same fused kernels with byte stores — both uint8 generations — and a float
    tensor that does not start on an element boundary is an error, not a silent change of path."""
    import ctypes

    from interpolate_antialiasing_amd import _lib, tables

    L = _lib.load()
    torch.manual_seed(21)
    x = torch.randint(0, 256, (3, 438, 906, 3), dtype=torch.uint8, device="cuda")  # NHWC bytes
    th = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_PIL, 438, 196, False, 0.0, x.device)
    tw = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_PIL, 906, 320, False, 0.0, x.device)
    ah, aw = th.axis(), tw.axis()
    assert L.aa_workspace_bytes(_lib.U8, _lib.NHWC, 3, 3, 438, 906, 196, 320, ctypes.byref(ah), ctypes.byref(aw)) == 0
    ref = aa.linear_forward(x.permute(0, 3, 1, 2), [196, 320]).permute(0, 2, 3, 1).contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for mode, want in ((1, "fused_u8_nhwc_pil_v3"), (2, "fused_u8_nhwc_pil")):
            _lib.set_fused(mode)
            for off in (1, 2, 3):
                buf = torch.zeros(ref.numel() + 8, dtype=torch.uint8, device="cuda")
                rc = L.aa_resample_fwd_ex(x.data_ptr(), buf.data_ptr() + off, None, 0, _lib.U8, _lib.NHWC, 3, 3, 438, 906,
                                          ctypes.byref(ah), ctypes.byref(aw), 0, stream)
                assert rc == 0, (mode, off, rc)
                assert _lib.last_variant() == want
                assert torch.equal(buf[off:off + ref.numel()], ref.reshape(-1)), (mode, off)
                assert int(buf[:off].sum()) == 0 and int(buf[off + ref.numel():].sum()) == 0  # nothing written outside the tensor
    finally:
        _lib.set_fused(1)
    xf = torch.rand(1, 1, 64, 64, device="cuda")
    tf = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_F32, 64, 32, False, 0.0, xf.device)
    af = tf.axis()
    out = torch.empty(32 * 32 + 4, device="cuda")
    rc = L.aa_resample_fwd_ex(xf.data_ptr(), out.data_ptr() + 2, None, 0, _lib.F32, _lib.NCHW, 1, 1, 64, 64, ctypes.byref(af), ctypes.byref(af), 0, stream)
    assert rc == -4, rc  # AA_ERR_BAD_SHAPE


def test_nonfinite_inputs_do_not_leak(aa):
    """fp32: a NaN/inf pixel may only reach the outputs whose window really holds it (zero-padded taps are not summed)."""
    x = torch.rand(1, 1, 64, 128, device="cuda")
    x[0, 0, 20, 50] = float("inf")
    x[0, 0, 40, 90] = float("nan")
    exp = oracle.forward("linear", x.cpu().numpy(), (23, 31))
    got = aa.linear_forward(x, [23, 31]).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(np.isinf(got), np.isinf(exp))
    ok = np.isfinite(exp)
    assert np.array_equal(got[ok], exp[ok])
    # growing heights take the gather-form kernel: same rule
    exp = oracle.forward("linear", x.cpu().numpy(), (100, 200))
    got = aa.linear_forward(x, [100, 200]).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(np.isinf(got), np.isinf(exp))
    ok = np.isfinite(exp)
    assert np.array_equal(got[ok], exp[ok])


# ------------------------------------------------------------------------------------------------ §8f: 16-bit floats, N-d
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("channels_last", [False, True])
def test_half_precision_is_rounded_fp32(aa, dtype, channels_last):
    """fp16/bf16 (not dispatched by the reference): defined as  half(reference_fp32(float(x)))  — fp32 arithmetic in
    the reference's order, fp32 intermediate, one round-to-nearest-even at the store.  Checked bit for bit against the
    oracle run on the up-cast input."""
    torch.manual_seed(11)
    for filt, shape, size in (("linear", (2, 3, 61, 53), (17, 23)), ("cubic", (1, 2, 40, 64), (13, 100)), ("box", (1, 1, 33, 35), (11, 7))):
        x = (torch.rand(shape) * 200 - 20).to(dtype)
        exp = torch.from_numpy(oracle.forward(filt, x.float().numpy(), size)).to(dtype)
        xg = x.cuda()
        if channels_last:
            xg = xg.contiguous(memory_format=torch.channels_last)
        got = _fn(aa, filt)(xg, list(size))
        assert got.dtype == dtype and got.is_contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        assert torch.equal(got.cpu().view(torch.int16), exp.view(torch.int16)), (filt, shape, size)
    if not channels_last:
        # the fused single-launch kernel (16-bit elements staged and read as halves, fp32 arithmetic, one rounding at the store)
        # at BASELINE sizes, odd widths included (rows that are only 2-byte aligned): bit-identical to the generic path, and to
        # the oracle on the up-cast input for one image
        from interpolate_antialiasing_amd import _lib

        want = "fused_f16_nchw" if dtype == torch.float16 else "fused_bf16_nchw"
        for fn, filt, shape, size in ((aa.linear_forward, "linear", (6, 3, 438, 906), [196, 320]), (aa.cubic_forward, "cubic", (2, 3, 438, 907), [196, 320]),
                                      (aa.cubic_forward, "cubic", (2, 2, 512, 512), [100, 160]), (aa.nearest_forward, "box", (1, 3, 333, 517), [100, 129])):
            x = ((torch.rand(shape, device="cuda") * 300) - 40).to(dtype)
            try:
                _lib.set_fused(1)
                y1 = fn(x, size)
                v = _lib.last_variant()
                _lib.set_fused(0)
                y0 = fn(x, size)
            finally:
                _lib.set_fused(1)
            assert v == want, (v, shape, size)
            assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), (filt, shape, size)
            exp1 = torch.from_numpy(oracle.forward(filt, x[:1].float().cpu().numpy(), tuple(size))).to(dtype)
            assert torch.equal(y1[:1].cpu().view(torch.int16), exp1.view(torch.int16)), (filt, shape, size)


def _oracle_axis(filt, a, axis, n_out, align_corners=False):
    """One separable pass along `axis` with the 2-D oracle: the axis becomes W of an [outer, inner, 1, n] image (the
    H pass is 1 -> 1, a single tap of weight exactly 1)."""
    moved = np.moveaxis(a, axis, -1)
    lead = moved.shape[:-1]
    img = np.ascontiguousarray(moved.reshape(-1, 1, 1, moved.shape[-1]))
    out = oracle.forward(filt, img, (1, n_out), align_corners)
    return np.moveaxis(out.reshape(*lead, n_out), -1, axis)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_1d_and_3d_front_ends(aa, dt):
    """NCL and NCDHW: the reference's driver resamples the LAST axis first (s2.2:658); the oracle is its
    2-D restatement applied axis by axis in that order, so results must be bit-identical."""
    rng = np.random.default_rng(5)
    x1 = (rng.random((3, 4, 97)) * 255).astype(dt)
    for filt, fn in (("linear", aa.linear_forward_nd), ("cubic", aa.cubic_forward_nd), ("box", aa.nearest_forward_nd)):
        for n_out in (31, 97, 150):
            got = fn(_gpu(x1), [n_out]).cpu().numpy()
            assert np.array_equal(got, _oracle_axis(filt, x1, 2, n_out)), (filt, n_out)
    x3 = (rng.random((2, 3, 19, 23, 29)) * 255).astype(dt)
    for filt, fn in (("linear", aa.linear_forward_nd), ("cubic", aa.cubic_forward_nd)):
        for size in ((7, 9, 11), (19, 40, 10), (30, 23, 29)):
            exp = x3
            for axis in (4, 3, 2):
                exp = _oracle_axis(filt, exp, axis, size[axis - 2])
            got = fn(_gpu(x3), list(size)).cpu().numpy()
            assert got.shape == (2, 3) + size and np.array_equal(got, exp), (filt, size)
    # align_corners only changes the scale (s2.2:314-315); 4-D input goes to the 2-D path; functional wrapper
    got = aa.linear_forward_nd(_gpu(x1), [40], True).cpu().numpy()
    assert np.array_equal(got, _oracle_axis("linear", x1, 2, 40, True))
    x2 = (rng.random((1, 2, 20, 30)) * 255).astype(dt)
    assert np.array_equal(aa.linear_forward_nd(_gpu(x2), [7, 9]).cpu().numpy(), oracle.forward("linear", x2, (7, 9)))
    from interpolate_antialiasing_amd.functional import interpolate_aa
    assert np.array_equal(interpolate_aa(_gpu(x3), (7, 9, 11), "trilinear").cpu().numpy(), aa.linear_forward_nd(_gpu(x3), [7, 9, 11]).cpu().numpy())


def test_nd_passes_take_the_fused_kernels(aa):
    """The 1-D / 3-D front-ends run every axis pass through the fused 2-D kernels (identity table on the other axis): at sizes
    where that matters the results stay bit-identical to the generic single-axis kernel and to the oracle applied axis by axis."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(41)
    x1 = torch.rand(4, 3, 4000, device="cuda") * 255 - 30
    x3 = torch.rand(2, 2, 40, 120, 300, device="cuda") * 255
    for fn, x, size in ((aa.linear_forward_nd, x1, [1300]), (aa.cubic_forward_nd, x1, [900]), (aa.linear_forward_nd, x1, [6000]),
                        (aa.linear_forward_nd, x3, [17, 50, 110]), (aa.cubic_forward_nd, x3, [40, 61, 300]), (aa.linear_forward_nd, x3, [60, 120, 100])):
        try:
            _lib.set_fused(1)
            y1 = fn(x, size)
            v = _lib.last_variant()
            _lib.set_fused(0)
            y0 = fn(x, size)
        finally:
            _lib.set_fused(1)
        assert v.startswith("fused"), (v, tuple(x.shape), size)
        assert torch.equal(y1, y0), (tuple(x.shape), size)
    exp = _oracle_axis("linear", x1[:1].cpu().numpy(), 2, 1300)
    assert np.array_equal(aa.linear_forward_nd(x1[:1], [1300]).cpu().numpy(), exp)
    # passes that would NOT be one fused launch go to the single-axis kernel (one launch, no workspace), never to the two-launch path
    # with an identity pass: the fused kernels switched off, fp64 with growing sizes (no fused kernel), rows too short for the
    # identity table to pay (round-2 advisor finding)
    xs = torch.rand(4, 3, 40, device="cuda")
    for fn, x, size, fused in ((aa.linear_forward_nd, x1, [1300], 0), (aa.linear_forward_nd, xs, [17], 1)):
        try:
            _lib.set_fused(fused)
            y = fn(x, size)
            v = _lib.last_variant()
        finally:
            _lib.set_fused(1)
        assert v == "generic_axis", (v, tuple(x.shape), size, fused)
        assert np.array_equal(y[:1].cpu().numpy(), _oracle_axis("linear", x[:1].cpu().numpy(), 2, size[0]))
    xd = torch.rand(1, 2, 40, 60, 70, device="cuda", dtype=torch.float64)  # the depth axis GROWS: fp64 has no fused kernel for growing heights
    yd = aa.linear_forward_nd(xd, [50, 60, 70])
    # (passes run last axis first: W and H keep their size = identity passes through the fused kernel, then the depth pass declines)
    assert _lib.last_variant() == "generic_axis", _lib.last_variant()
    assert np.array_equal(yd.cpu().numpy(), _oracle_axis("linear", xd.cpu().numpy(), 2, 50))
    g = torch.randn(4, 3, 1300, device="cuda", dtype=torch.float64)
    try:
        _lib.set_fused(1)
        b1 = aa.linear_backward_nd(g, [1300], [4, 3, 4000])
        _lib.set_fused(0)
        b0 = aa.linear_backward_nd(g, [1300], [4, 3, 4000])
    finally:
        _lib.set_fused(1)
    assert torch.equal(b1, b0)


def test_nd_matches_torch_upstream(aa):
    """Independent cross-check (never the implementation): PyTorch's own antialiased interpolate is 2-D only, so compare
    the 1-D front-end with F.interpolate on an [N,C,1,L] view (its H pass is an exact identity)."""
    import torch.nn.functional as F

    x = torch.rand(2, 3, 211, device="cuda", dtype=torch.float64)
    got = aa.linear_forward_nd(x, [64])
    ref = F.interpolate(x[:, :, None, :], size=(1, 64), mode="bilinear", antialias=True, align_corners=False)[:, :, 0, :]
    assert torch.allclose(got, ref, rtol=0, atol=1e-12)


def test_rccl_broadcast_path_single_rank(aa):
    """The multi-GPU path's only collective (rank 0's packed tables -> everyone, then cached) through the REAL backend:
    a one-rank RCCL group on this box exercises the calls bench.py makes at N > 1 (group creation bound to the device,
    descriptor + payload broadcasts of GPU buffers, the MAX/SUM reductions).  World-size-2 logic runs under gloo in the
    CPU suite."""
    import socket

    import torch.distributed as dist
    from interpolate_antialiasing_amd import _lib, sharding, tables

    if dist.is_initialized():
        pytest.skip("a process group already exists")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        t = tables.get_table(_lib.FILTER_LINEAR, _lib.TABLE_PIL, 906, 320, False, 0.0, dev)
        r = sharding.broadcast_table(t, src=0, device=dev)
        assert r is t
        v = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        n = torch.tensor([7], dtype=torch.int64, device=dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        dist.barrier()
        assert float(v.item()) == 1.5 and int(n.item()) == 7
        th, tw = sharding.prepare_tables(_lib.FILTER_LINEAR, _lib.TABLE_PIL, (438, 906), (196, 320), False, dev)
        assert (th.in_size, th.out_size, tw.in_size, tw.out_size) == (438, 196, 906, 320)
    finally:
        dist.destroy_process_group()


def test_explicit_scale_factors(aa):
    """ATen's optional scale factors (the reference's binding hard-wires none, s2.2:11-13): a given factor s
    makes the scale 1/s instead of in/out (area_pixel_compute_scale), which moves windows and weights.  Tables bit-exact
    against the oracle's restatement of that rule; the forward result against those tables applied in the reference's
    order (tap 0 first, product and sum rounded separately), W pass then H pass."""
    from interpolate_antialiasing_amd import _lib, tables

    rng = np.random.default_rng(9)
    x = (rng.random((2, 3, 50, 70)) * 255).astype(np.float32)
    oh, ow, sh, sw = 21, 31, 0.4, 0.45  # floor(50 * 0.4) = 20 != 21: the factor, not the size ratio, sets the scale
    for filt in ("linear", "cubic"):
        tabs = []
        for n_in, n_out, s in ((50, oh, sh), (70, ow, sw)):
            k, xmin, xsize, w = oracle.weights(filt, n_in, n_out, False, np.float32, scale=s)
            t = tables.build_table(oracle.FILTERS[filt], _lib.TABLE_F32, n_in, n_out, False, s, torch.device("cuda"))
            dxmin, dxsize, dw = t.unpack()
            assert t.ksize == k and np.array_equal(dxmin, xmin) and np.array_equal(dxsize, xsize) and np.array_equal(dw, w)
            k0, xmin0, _, _ = oracle.weights(filt, n_in, n_out, False, np.float32)
            assert k0 != k or not np.array_equal(xmin0, xmin)  # the factor really changes the table
            tabs.append((xmin, xsize, w))

        def apply(a, tab, n_out):  # along the last axis
            xmin, xsize, w = tab
            out = np.empty(a.shape[:-1] + (n_out,), np.float32)
            for o in range(n_out):
                acc = a[..., xmin[o]] * w[o, 0]
                for j in range(1, max(int(xsize[o]), 1)):
                    acc = acc + a[..., xmin[o] + j] * w[o, j]
                out[..., o] = acc
            return out

        mid = apply(x, tabs[1], ow)                                # W pass first (s2.2:658)
        exp = np.swapaxes(apply(np.swapaxes(mid, 2, 3), tabs[0], oh), 2, 3)
        got = _fn(aa, filt)(_gpu(x), [oh, ow], False, scale_factors=[sh, sw]).cpu().numpy()
        assert np.array_equal(got, exp), filt
    with pytest.raises(NotImplementedError, match="no scale factors"):
        aa.linear_forward(torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device="cuda"), [4, 4], scale_factors=[0.5, 0.5])


def test_nd_backward_is_the_adjoint(aa):
    """1-D / 3-D backward: <A x, y> == <x, A^T y> in fp64 for the front-ends' forward A (size-independent property), and
    the 1-D case against autograd of PyTorch's own antialiased interpolate on an [N,C,1,L] view."""
    import torch.nn.functional as F

    torch.manual_seed(6)
    for fwd, bwd, shape, size in ((aa.linear_forward_nd, aa.linear_backward_nd, (2, 3, 57), (19,)),
                                  (aa.cubic_forward_nd, aa.cubic_backward_nd, (1, 2, 40), (90,)),
                                  (aa.linear_forward_nd, aa.linear_backward_nd, (2, 2, 11, 13, 17), (5, 20, 7)),
                                  (aa.cubic_forward_nd, aa.cubic_backward_nd, (1, 1, 9, 8, 30), (9, 3, 11))):
        x = torch.randn(shape, dtype=torch.float64, device="cuda")
        y = torch.randn(shape[:2] + size, dtype=torch.float64, device="cuda")
        lhs = (fwd(x, list(size)) * y).sum()
        rhs = (x * bwd(y, list(size), list(shape))).sum()
        assert abs(lhs.item() - rhs.item()) <= 1e-10 * max(1.0, abs(lhs.item())), (shape, size)
    x = torch.randn(2, 3, 211, dtype=torch.float64, device="cuda", requires_grad=True)
    g = torch.randn(2, 3, 64, dtype=torch.float64, device="cuda")
    ref = F.interpolate(x[:, :, None, :], size=(1, 64), mode="bilinear", antialias=True, align_corners=False)[:, :, 0, :]
    ref.backward(g)
    got = aa.linear_backward_nd(g, [64], [2, 3, 211])
    assert torch.allclose(got, x.grad, rtol=0, atol=1e-12)


def test_hip_graph_capture_and_replay(aa):
    """The launches are stream-ordered and allocation-free once the tables are cached, so a latency-bound caller can
    capture them in a HIP graph (torch.cuda.CUDAGraph) and replay: same bytes as the eager call, for new input data."""
    torch.manual_seed(12)
    xs = torch.randint(0, 256, (2, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    gs = torch.randn(2, 3, 196, 320, device="cuda")
    aa.linear_forward(xs, [196, 320])                      # builds and caches the tables (the only synchronising step)
    aa.linear_backward(gs, [196, 320], [2, 3, 438, 906])
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                          # warm-up on the capture stream, as torch requires
        aa.linear_forward(xs, [196, 320])
        aa.linear_backward(gs, [196, 320], [2, 3, 438, 906])
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = aa.linear_forward(xs, [196, 320])
        gi = aa.linear_backward(gs, [196, 320], [2, 3, 438, 906])
    for seed in (1, 2):
        torch.manual_seed(seed)
        xs.copy_(torch.randint(0, 256, (2, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2))
        gs.copy_(torch.randn(2, 3, 196, 320, device="cuda"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, aa.linear_forward(xs, [196, 320]))
        assert torch.equal(gi, aa.linear_backward(gs, [196, 320], [2, 3, 438, 906]))


# ------------------------------------------------------------------------------------------------ §8f-1: the reference's harness
@pytest.mark.parametrize("mode,filt", [("bilinear", "linear"), ("bicubic", "cubic")])
def test_harness_five_sizes_vs_pil(aa, golden_kat, golden_harness, mode, filt):
    """The reference's own check (test.py:334-379) on its own image at its five (W, H) sizes (test.py:15-21), through
    tools/harness.py's logic: test.py's flow (float(), op, clamp for bicubic, byte()) must meet test.py's thresholds against
    PIL (MAE < 1, max < 1 + 1e-5 bilinear / < 20 bicubic, :370-379) and equal the oracle's harness restatement bit for bit;
    the Pillow-exact mode must equal PIL's committed output bit for bit.  Includes both up-scaling sizes ((1200, 196) widens,
    (120, 1200) heightens: the uint8 kernels' gather-form vertical pass)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import harness

    rgb = golden_kat["rgb"]
    chw = np.ascontiguousarray(rgb.transpose(2, 0, 1))[None]
    fn = _fn(aa, filt)
    assert harness.SIZES == [(320, 196), (460, 220), (120, 96), (1200, 196), (120, 1200)]
    for (w, h) in harness.SIZES:
        pil = golden_harness[f"pil_{filt}_{w}x{h}"]
        r = harness.evaluate(rgb, (w, h), mode, "harness", pil_dn=pil)
        assert r["mae"] < harness.THRESHOLDS[mode][0] and r["max"] < harness.THRESHOLDS[mode][1], (w, h, r["mae"], r["max"])
        assert abs(r["mae"] - float(golden_harness[f"refmae_{filt}_{w}x{h}"])) < 1e-12, (w, h)  # the reference build's own MAE
        exp = oracle.harness_u8(filt, chw, (h, w))[0].transpose(1, 2, 0)
        assert np.array_equal(r["proto"], exp), (w, h, r["variant"])
        # the shim's fused form of the same semantics (uint8 in, uint8 out) in both layouts
        for cl in (False, True):
            y = fn(_gpu(chw, cl), [h, w], False, uint8_mode="harness")
            assert np.array_equal(y[0].permute(1, 2, 0).cpu().numpy(), exp), (w, h, cl)
        p = harness.evaluate(rgb, (w, h), mode, "pil", pil_dn=pil)
        assert p["max"] == 0.0 and np.array_equal(p["proto"], pil), (w, h, p["variant"])
        ycl = fn(_gpu(chw, True), [h, w], False)  # channels_last = what PIL itself holds (HWC)
        assert np.array_equal(ycl[0].permute(1, 2, 0).cpu().numpy(), pil), (w, h)


def test_fused_segments_follow_the_tables_scale(aa):
    """Staged row segments are sized from the spread of the table's own window starts (header.span64p1), not from W/oW:
    an explicit scale factor with 1/s > W/oW, or align_corners when down-scaling, spreads 64 consecutive windows further
    apart than 63*W/oW.  Fused == generic bit for bit, with oW >= 64 so that a strip really spans 64 outputs."""
    from interpolate_antialiasing_amd import _lib, tables

    torch.manual_seed(11)
    xf = torch.rand(2, 3, 120, 906, device="cuda") * 255
    x8 = torch.randint(0, 256, (2, 120, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    t0 = tables.build_table(_lib.FILTER_LINEAR, _lib.TABLE_F32, 906, 320, False, 0.0, torch.device("cuda"))
    t1 = tables.build_table(_lib.FILTER_LINEAR, _lib.TABLE_F32, 906, 320, False, 0.30, torch.device("cuda"))
    xmin1 = t1.unpack()[0]
    assert t1.span64p1 == 1 + int((xmin1[63:] - xmin1[:-63]).max()) and t1.span64p1 > t0.span64p1 + 20
    calls = [
        (lambda: aa.linear_forward(xf, [50, 320], scale_factors=[0.0, 0.30]), "fused_f32_nchw"),        # 1/0.30 = 3.33 > 2.83
        (lambda: aa.cubic_forward(xf, [50, 320], scale_factors=[0.0, 0.30]), "fused_f32_nchw"),
        (lambda: aa.linear_forward(x8, [50, 320], uint8_mode="harness", scale_factors=[0.0, 0.30]), "fused_u8_nhwc_harness_v3"),
        (lambda: aa.linear_forward(x8, [50, 320], True, uint8_mode="harness"), "fused_u8_nhwc_harness_v3"),  # align_corners
        (lambda: aa.linear_forward(xf, [50, 320], True), "fused_f32_nchw"),                               # (W-1)/(oW-1) > W/oW
        (lambda: aa.linear_forward(xf, [300, 320], scale_factors=[0.0, 0.30]), "fused_f32_nchw_up"),     # heights grow
    ]
    try:
        for fn, want in calls:
            _lib.set_fused(1)
            y1 = fn()
            assert _lib.last_variant() == want, (_lib.last_variant(), want)
            _lib.set_fused(0)
            y0 = fn()
            assert _lib.last_variant().startswith("generic")
            assert torch.equal(y1, y0), want
    finally:
        _lib.set_fused(1)


def test_workspace_answer_matches_dispatch(aa):
    """aa_workspace_bytes() says 0 exactly when aa_resample_fwd() will not ask for one: thumbnail shapes whose windows (13-16
    taps) are too wide for the newest uint8 kernel and whose intermediate ring is too big for the first one must run the
    generic path with a workspace instead of failing (uint8 channels_last bicubic 1750 -> 500, bilinear 2400 -> 400)."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(12)
    for fn, shape, size, want in ((aa.cubic_forward, (2, 1750, 1750, 3), [500, 500], "fused_u8_nhwc_pil_v3"),     # 15 taps
                                  (aa.linear_forward, (1, 300, 2400, 3), [50, 400], "fused_u8_nhwc_pil_v3"),      # 13 taps
                                  (aa.linear_forward, (1, 2400, 2400, 3), [400, 400], "fused_u8_nhwc_pil_v3"),
                                  (aa.cubic_forward, (1, 700, 1400, 4), [100, 200], "fused_u8_nhwc_pil_v3"),     # 29 taps: the 34-tap window (round 3)
                                  (aa.cubic_forward, (1, 700, 1400, 4), [100, 100], "fused_u8_nhwc_pil_v3"),     # 57 taps: split windows, four lanes per pixel (round 3)
                                  (aa.cubic_forward, (1, 700, 1400, 4), [100, 38], "generic_2pass_u8_pil")):      # 149 taps: no fused kernel
        x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
        y = fn(x, size)
        v = _lib.last_variant()
        try:
            _lib.set_fused(0)
            y0 = fn(x, size)
        finally:
            _lib.set_fused(1)
        assert torch.equal(y, y0), (shape, size, v)
        assert v == want, (v, shape, size)  # 13-16 taps: the 16-tap window instantiation of the newest kernel
        exp = oracle.pil_resize_u8("cubic" if fn is aa.cubic_forward else "linear", x[:1].cpu().numpy(), tuple(size))
        assert np.array_equal(y[:1].cpu().numpy(), exp), (shape, size, v)


# ------------------------------------------------------------------------------------------------ §8f-3: decode-adjacent fusion
def test_u8_to_float_conversion_fused(aa, golden_kat):
    """uint8 HWC / CHW in -> float32 out in ONE launch (test.py:337-339,55: np.asarray(pil) -> transpose -> .float() -> op):
    bit-identical to the oracle's fp32 forward on the converted image (= the harness arithmetic before its byte()), at the
    BASELINE config-1 size on the reference's own image, in every layout pair, with and without (v - mean) / std."""
    from interpolate_antialiasing_amd import _lib

    rgb = golden_kat["rgb"]                                     # [438, 906, 3] uint8, as PIL holds it
    chw = np.ascontiguousarray(rgb.transpose(2, 0, 1))[None]
    exp = {f: oracle.forward(f, chw.astype(np.float32), (196, 320)) for f in ("linear", "cubic")}
    assert np.array_equal(exp["linear"], golden_kat["lin_320x196_f32"])   # = the reference build's own output
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    m32, s32 = np.asarray(mean, np.float32).reshape(1, 3, 1, 1), np.asarray(std, np.float32).reshape(1, 3, 1, 1)
    hwc = torch.from_numpy(rgb.copy()).cuda()[None].permute(0, 3, 1, 2)  # channels_last view of the HWC bytes
    planar = torch.from_numpy(chw).cuda()
    seen = set()
    for filt in ("linear", "cubic"):
        fn = _fn(aa, filt)
        for x, fmts in ((hwc, ("nchw", "nhwc", None)), (planar, ("nchw", None, "nhwc"))):
            for fmt in fmts:
                y = fn(x, [196, 320], out_dtype=torch.float32, out_format=fmt)
                seen.add(_lib.last_variant())
                assert y.dtype == torch.float32 and tuple(y.shape) == (1, 3, 196, 320)
                want_cl = (fmt == "nhwc") or (fmt is None and x is hwc)
                assert y.is_contiguous(memory_format=torch.channels_last if want_cl else torch.contiguous_format)
                assert np.array_equal(y.cpu().numpy(), exp[filt]), (filt, fmt, _lib.last_variant())
                yn = fn(x, [196, 320], out_dtype=torch.float32, out_format=fmt, mean=mean, std=std)
                assert np.array_equal(yn.cpu().numpy(), (exp[filt] - m32) / s32), (filt, fmt, "normalised")
    assert {"fused_u8_nhwc_to_f32_nchw_v3", "fused_u8_nhwc_to_f32_nhwc_v3", "fused_u8_planar_to_f32_v3",
            "generic_2pass_u8_to_f32"} <= seen, seen
    # batched, 4 channels, up-scaling (generic form), fused == generic
    torch.manual_seed(21)
    x4 = torch.randint(0, 256, (3, 200, 301, 4), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    for size in ([77, 100], [300, 100], [64, 512]):
        try:
            _lib.set_fused(1)
            y1 = aa.linear_forward(x4, size, out_dtype=torch.float32, out_format="nchw", mean=[1, 2, 3, 4], std=[2, 3, 4, 5])
            _lib.set_fused(0)
            y0 = aa.linear_forward(x4, size, out_dtype=torch.float32, out_format="nchw", mean=[1, 2, 3, 4], std=[2, 3, 4, 5])
        finally:
            _lib.set_fused(1)
        assert torch.equal(y1, y0), size
        ref = oracle.forward("linear", x4[:1].cpu().numpy().astype(np.float32), tuple(size))
        refn = (ref - np.asarray([1, 2, 3, 4], np.float32).reshape(1, 4, 1, 1)) / np.asarray([2, 3, 4, 5], np.float32).reshape(1, 4, 1, 1)
        assert np.array_equal(y1[:1].cpu().numpy(), refn), size
    with pytest.raises(NotImplementedError):
        aa.linear_forward(hwc.float(), [196, 320], out_dtype=torch.float32)
    with pytest.raises(NotImplementedError):
        aa.linear_forward(hwc, [196, 320], out_dtype=torch.float32, uint8_mode="pil")


# ------------------------------------------------------------------------------------------------ the reference-side binding
def test_integration_stub_matches_the_shim(aa):
    """INTEGRATION.md's pybind11 module (tools/integration_stub/extension_interpolate_amd.cpp: the reference's four callables
    with their bodies replaced by C-ABI calls) really compiles against the installed PyTorch and gives the shim's results bit
    for bit: uint8 channels_last (Pillow-exact), fp32 NCHW and channels_last, bicubic, box, and the true-adjoint backward."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "integration_stub", "build.py")
    spec = importlib.util.spec_from_file_location("aa_stub_build", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stub = mod.build()
    torch.manual_seed(31)
    x8 = torch.randint(0, 256, (3, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    xf = torch.rand(2, 3, 438, 906, device="cuda") * 255
    for x in (x8, xf, xf.contiguous(memory_format=torch.channels_last), xf.double()[:1]):
        for name in ("linear_forward", "cubic_forward", "nearest_forward"):
            y = getattr(stub, name)(x, [196, 320])
            z = getattr(aa, name)(x, [196, 320])
            assert y.dtype == z.dtype and y.stride() == z.stride() and torch.equal(y, z), (name, x.dtype)
    exp = oracle.pil_resize_u8("linear", x8[:1].cpu().numpy(), (196, 320))
    assert np.array_equal(stub.linear_forward(x8[:1], [196, 320], False).cpu().numpy(), exp)
    g = torch.randn(2, 3, 196, 320, device="cuda")
    assert torch.equal(stub.linear_backward(g, [196, 320], [2, 3, 438, 906], False), aa.linear_backward(g, [196, 320], [2, 3, 438, 906]))
    with pytest.raises(RuntimeError, match="It is expected output_size equals to 2"):
        stub.linear_forward(xf, [4, 4, 4])


def test_plane_groups_equal_single_planes(aa):
    """Planar (NCHW) uint8 images of three channels run their three planes in one wave (template parameter PL of the fused kernel);
    aa_set_plane_groups(0) restores one wave per plane.  Both forms and the two-launch path must agree bit for bit: ragged widths (byte
    stores), odd plane sizes (every plane on its own 16-byte phase), negative weights, several strips, a cropped view, and the
    oracle on a small case."""
    from interpolate_antialiasing_amd import _lib

    cases = [((5, 3, 90, 438), (40, 196), "linear"), ((2, 3, 131, 307), (37, 101), "linear"), ((3, 3, 70, 301), (20, 133), "cubic"),
             ((2, 3, 64, 1000), (31, 420), "linear"), ((1, 3, 33, 97), (9, 16), "box"), ((4, 3, 57, 83), (19, 27), "linear")]
    try:
        for shape, out, filt in cases:
            x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")
            fn = _fn(aa, filt)
            _lib.set_plane_groups(1)
            y1 = fn(x, list(out))
            v1 = _lib.last_variant()
            _lib.set_plane_groups(0)
            y2 = fn(x, list(out))
            v2 = _lib.last_variant()
            _lib.set_fused(0)
            y0 = fn(x, list(out))
            _lib.set_fused(1)
            assert v1 == v2 == "fused_u8_planar_pil_v3", (shape, out, v1, v2)
            assert torch.equal(y1, y0), (shape, out, filt, "plane groups")
            assert torch.equal(y2, y0), (shape, out, filt, "single planes")
        # float arithmetic (test.py's own path: CHW bytes, float(), op, byte()) and float32 planes out, exact and in the tolerance mode
        for shape, out, filt in (((3, 3, 90, 438), (40, 196), "linear"), ((2, 3, 131, 307), (37, 101), "linear"), ((2, 3, 70, 301), (31, 150), "cubic")):
            x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")
            fn = _fn(aa, filt)
            for kw, want in (({"uint8_mode": "harness"}, "fused_u8_planar_harness_v3"),
                             ({"out_dtype": torch.float32, "mean": [1.0, 2.0, 3.0], "std": [2.0, 3.0, 4.0]}, "fused_u8_planar_to_f32_v3")):
                _lib.set_plane_groups(1)
                y1 = fn(x, list(out), **kw)
                v1 = _lib.last_variant()
                yf = fn(x, list(out), precision="fast", **kw)
                vf = _lib.last_variant()
                _lib.set_plane_groups(0)
                y2 = fn(x, list(out), **kw)
                _lib.set_fused(0)
                y0 = fn(x, list(out), **kw)
                _lib.set_fused(1)
                assert v1 == want and vf == want + "_fast", (shape, out, v1, vf)
                assert torch.equal(y1, y0) and torch.equal(y2, y0), (shape, out, filt, want)
                if y1.dtype == torch.uint8:
                    assert (yf.int() - y0.int()).abs().max().item() <= 1
                else:
                    torch.testing.assert_close(yf, y0, rtol=1e-4, atol=1e-3)
        # groups are three CONSECUTIVE planes of the tensor, whatever image they belong to: grayscale batches, 2 / 4 / 5 channels, plane
        # counts that leave a last group of one or two (whose missing planes are neither read past the tensor nor stored)
        for shape, out in (((7, 1, 90, 438), (40, 196)), ((4, 1, 64, 300), (30, 132)), ((3, 2, 57, 83), (19, 27)), ((2, 4, 70, 301), (33, 140)),
                           ((1, 5, 131, 307), (37, 101)), ((1, 1, 50, 60), (20, 24)), ((2, 1, 45, 77), (17, 30))):
            x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")
            for kw in ({}, {"uint8_mode": "harness"}, {"out_dtype": torch.float32, "mean": [1.0] * shape[1], "std": [2.0, 3.0, 4.0, 5.0, 6.0][:shape[1]]}):
                if "mean" in kw and shape[1] > 4:
                    continue
                _lib.set_plane_groups(1)
                y1 = aa.linear_forward(x, list(out), **kw)
                _lib.set_plane_groups(0)
                y2 = aa.linear_forward(x, list(out), **kw)
                _lib.set_fused(0)
                y0 = aa.linear_forward(x, list(out), **kw)
                _lib.set_fused(1)
                assert torch.equal(y1, y0) and torch.equal(y2, y0), (shape, out, kw)
        _lib.set_plane_groups(1)
        x = torch.randint(0, 256, (2, 3, 45, 77), dtype=torch.uint8, device="cuda")
        y = aa.linear_forward(x, [17, 30])
        assert np.array_equal(y.cpu().numpy(), oracle.pil_resize_u8("linear", x.cpu().numpy(), (17, 30)))
        big = torch.randint(0, 256, (4, 3, 80, 330), dtype=torch.uint8, device="cuda")
        view = big[1:3, :, 5:75, 13:313]
        yv = aa.linear_forward(view, [30, 132])
        assert _lib.last_variant() == "fused_u8_planar_pil_v3"
        assert torch.equal(yv, aa.linear_forward(view.contiguous(), [30, 132]))
    finally:
        _lib.set_fused(1)
        _lib.set_plane_groups(1)


def test_fp32_channels_last_strong_downscale_is_fused(aa):
    """fp32 channels_last at test.py's 906 -> 120 thumbnail width (17 bilinear taps): strips of 32 elements keep the staged segment
    within the kernel's 128 pieces, so the shape runs fused (round 3; it was on the two-launch path); bit-identical to that path and to
    the oracle."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(5)
    for c in (3, 4):
        x = (torch.rand(2, c, 438, 906, device="cuda") * 255).contiguous(memory_format=torch.channels_last)
        y = aa.linear_forward(x, [96, 120])
        assert _lib.last_variant() == "fused_f32_nhwc", (c, _lib.last_variant())
        try:
            _lib.set_fused(0)
            y0 = aa.linear_forward(x, [96, 120])
        finally:
            _lib.set_fused(1)
        assert torch.equal(y, y0), c
        exp = oracle.forward("linear", x[:1].contiguous().cpu().numpy(), (96, 120))
        assert np.array_equal(y[:1].contiguous().cpu().numpy(), exp), c
        # bicubic: 33 taps (9 quads of window positions, vector-register lane masks), and 25 taps (7 quads)
        for size in ([96, 120], [110, 160]):
            y = aa.cubic_forward(x, size)
            assert _lib.last_variant() == "fused_f32_nhwc", (c, size, _lib.last_variant())
            try:
                _lib.set_fused(0)
                y0 = aa.cubic_forward(x, size)
            finally:
                _lib.set_fused(1)
            assert torch.equal(y, y0), (c, size)


def test_sixteen_bit_wide_windows_are_fused(aa):
    """fp16 / bf16 planes with 18 .. 33 taps (test.py's bicubic 906 -> 120 thumbnails; config 2's bicubic 1024 -> 224 in halves): five quads
    of eight window positions, lane masks in vector registers.  Bit-identical to the two-launch path and to half(oracle_fp32(float(x)));
    the tolerance mode within its bar."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(6)
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for shape, size in (((2, 3, 438, 906), [96, 120]), ((2, 3, 1024, 1024), [224, 224]), ((1, 2, 300, 701), [70, 100])):
            x = (torch.rand(*shape, device="cuda") * 255).to(dt)
            y = aa.cubic_forward(x, size)
            assert _lib.last_variant() == f"fused_{name}_nchw", (name, shape, _lib.last_variant())
            yf = aa.cubic_forward(x, size, precision="fast")
            assert _lib.last_variant() == f"fused_{name}_nchw_fast", _lib.last_variant()
            try:
                _lib.set_fused(0)
                y0 = aa.cubic_forward(x, size)
            finally:
                _lib.set_fused(1)
            assert torch.equal(y.view(torch.int16), y0.view(torch.int16)), (name, shape)
            torch.testing.assert_close(yf.float(), y0.float(), rtol=2e-2, atol=2.0)
        x = (torch.rand(1, 1, 120, 400, device="cuda") * 255).to(dt)
        y = aa.cubic_forward(x, [30, 55])
        exp = torch.from_numpy(oracle.forward("cubic", x.float().cpu().numpy(), (30, 55))).to(dt)
        assert torch.equal(y.cpu().view(torch.int16), exp.view(torch.int16)), name


def test_fp16_products_are_the_references(aa):
    """The fp16 kernel multiplies with v_fma_mix_f32 (half operand taken straight from the packed register, addend -0.0): the product must
    be the separately rounded float(h) * w of the reference in every corner — denormal halves, signed zeros (a zero product keeps its sign,
    so an all-zero window gives the reference's zero), large magnitudes, negative bicubic weights — bit-identical to the two-launch path,
    which converts with v_cvt_f32_f16 and multiplies."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(9)
    x = (torch.rand(2, 3, 200, 333, device="cuda") * 255).half()
    x[0, 0, :50] = 0.0
    x[0, 1, :50] = -0.0
    x[0, 2, 10:60, 20:200] = torch.tensor(6.0e-8, dtype=torch.float16, device="cuda")      # the smallest denormal half
    x[1, 0, 100:150] = (torch.rand(50, 333, device="cuda") * 6.0e-5).half()               # denormals and tiny normals
    x[1, 1, 30:90, 50:300] = -(torch.rand(60, 250, device="cuda") * 6.0e4).half()         # large negatives
    x[1, 2, ::3] = 65504.0
    for fn, size in ((aa.linear_forward, [90, 150]), (aa.cubic_forward, [90, 150]), (aa.cubic_forward, [40, 44]), (aa.linear_forward, [200, 120])):
        y = fn(x, size)
        assert _lib.last_variant() == "fused_f16_nchw", _lib.last_variant()
        try:
            _lib.set_fused(0)
            y0 = fn(x, size)
        finally:
            _lib.set_fused(1)
        assert torch.equal(y.view(torch.int16), y0.view(torch.int16)), size   # bit patterns: -0.0 and +0.0 are told apart
        yf = fn(x, size, precision="fast")
        assert _lib.last_variant() == "fused_f16_nchw_fast"
        ok = torch.isfinite(y0.float())
        torch.testing.assert_close(yf.float()[ok], y0.float()[ok], rtol=2e-3, atol=1e-3)


def test_sixteen_bit_growing_heights_are_fused(aa):
    """fp16 / bf16 planes whose height does not shrink (test.py's 1200 x 1200 and (1200, 196)-like sizes in halves): the up-scaling kernel
    stages and reads halves, computes in fp32 and rounds once at the store (round 3; these shapes ran two launches).  Bit-identical to the
    two-launch path and to half(oracle_fp32(float(x))): 4 / 2 / 1 columns per lane, ragged widths, odd W (the tensor's final element is
    fetched on its own: nothing is read past the tensor), plain and streaming store forms."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(8)
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for shape, size, filt in (((2, 3, 64, 120), (150, 300), "linear"), ((2, 3, 64, 121), (150, 301), "cubic"), ((1, 2, 33, 19), (70, 50), "linear"),
                                  ((3, 1, 40, 57), (44, 130), "cubic"), ((2, 3, 100, 200), (240, 203), "linear"), ((1, 3, 438, 906), (1200, 1200), "linear"),
                                  ((2, 4, 32, 512), (96, 512), "linear")):   # (a 19 x 512-byte-class tensor: ends on its allocation)
            x = ((torch.rand(*shape, device="cuda") * 300) - 40).to(dt)
            fn = _fn(aa, filt)
            for form in (-1, 1):
                prev = _lib.set_store_form(form)
                try:
                    y = fn(x, list(size))
                    v = _lib.last_variant()
                    _lib.set_fused(0)
                    y0 = fn(x, list(size))
                finally:
                    _lib.set_fused(1)
                    _lib.set_store_form(prev)
                assert v == f"fused_{name}_nchw_up", (name, shape, size, v)
                assert torch.equal(y.view(torch.int16), y0.view(torch.int16)), (name, shape, size, filt, form)
        x = ((torch.rand(1, 2, 30, 41, device="cuda") * 300) - 40).to(dt)
        y = aa.linear_forward(x, [70, 90])
        exp = torch.from_numpy(oracle.forward("linear", x.float().cpu().numpy(), (70, 90))).to(dt)
        assert torch.equal(y.cpu().view(torch.int16), exp.view(torch.int16)), name


def test_sixteen_bit_tensor_ending_on_its_allocation(aa):
    """Rows of 16-bit elements with an odd W: the dword that holds the tensor's final element straddles the end of the tensor.  The
    fused kernel must neither drop that element nor read the two bytes beyond it (round 3: the second form faulted once in 90 000
    fuzz problems, when the tensor ended on the last byte of a mapped block).  The tensors here fill their allocation exactly:
    18 MiB (a block of its own from the caching allocator) and 19 x 512 bytes (the fuzz case)."""
    from interpolate_antialiasing_amd import _lib

    for dt in (torch.bfloat16, torch.float16):
        for shape, out in (((16, 64, 1024, 9), (512, 4)), ((2, 4, 32, 19), (11, 4)), ((1, 1, 33, 19), (33, 7))):
            x = ((torch.rand(*shape, device="cuda") * 300) - 40).to(dt)
            try:
                _lib.set_fused(1)
                y1 = aa.linear_forward(x, list(out))
                v = _lib.last_variant()
                yf = aa.linear_forward(x, list(out), precision="fast")
                _lib.set_fused(0)
                y0 = aa.linear_forward(x, list(out))
            finally:
                _lib.set_fused(1)
            torch.cuda.synchronize()
            assert v.startswith("fused_"), (shape, v)
            assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), (dt, shape)
            assert torch.isfinite(yf.float()).all()
            torch.testing.assert_close(yf.float(), y0.float(), rtol=2e-2, atol=2.0)  # (16-bit outputs: one rounding step apart at most)
            del x, y0, y1, yf


# ------------------------------------------------------------------------------------------------ fuzz: fused == generic
def test_fuzz_fused_equals_generic(aa):
    """120 seeded random problems (dtype, layout, channels, sizes from 1 to ~700, down / up / mixed scales, three filters,
    uint8 in both arithmetics, uint8 -> float32 conversion, backward): whatever kernel the dispatcher picks must agree bit for bit
    with the generic two-launch path.  Catches edge cases of the strip / segment / window-alignment logic that fixed shapes miss.
    (Round 2 also ran this loop as a soak with the final kernels: 5 other seeds x 20 000 problems, all 100 000 bit-identical, about
    half of them on fused kernels.)"""
    from interpolate_antialiasing_amd import _lib

    import os

    cases = int(os.environ.get("AA_FUZZ_CASES", "120"))  # (a long soak: AA_FUZZ_CASES=3000 AA_FUZZ_SEED=<n>)
    rng = np.random.default_rng(int(os.environ.get("AA_FUZZ_SEED", "20260502")))
    dtypes = [torch.uint8, torch.uint8, torch.float32, torch.float32, torch.float64, torch.float16, torch.bfloat16]
    fused = 0
    for it in range(cases):
        dt = dtypes[int(rng.integers(len(dtypes)))]
        c = int(rng.choice([1, 2, 3, 3, 4, 5]))
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        def pick(v):
            r = rng.random()
            if r < 0.55: return max(1, int(v / rng.uniform(1.0, 6.0)))
            if r < 0.8: return min(900, max(1, int(v * rng.uniform(1.0, 3.0))))
            return v
        oh, ow = pick(h), pick(w)
        cl = bool(rng.integers(2))
        filt = ["linear", "cubic", "box"][int(rng.integers(3))]
        fn = _fn(aa, filt)
        if dt == torch.uint8:
            x = torch.randint(0, 256, (n, c, h, w), dtype=torch.uint8, device="cuda")
        else:
            x = ((torch.rand(n, c, h, w, device="cuda") * 300) - 40).to(dt)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        kind = int(rng.integers(4))
        if dt == torch.uint8 and kind == 1:
            call = lambda: fn(x, [oh, ow], uint8_mode="harness")
        elif dt == torch.uint8 and kind == 2 and c <= 4:
            call = lambda: fn(x, [oh, ow], out_dtype=torch.float32, out_format=["nchw", "nhwc"][int(it % 2)], mean=[1.0] * c, std=[2.0] * c)
        elif dt in (torch.float32, torch.float64) and kind == 3 and filt != "box":
            g = torch.randn(n, c, oh, ow, device="cuda", dtype=dt)
            if cl:
                g = g.contiguous(memory_format=torch.channels_last)
            bw = aa.linear_backward if filt == "linear" else aa.cubic_backward
            call = lambda: bw(g, [oh, ow], [n, c, h, w])
        else:
            call = lambda: fn(x, [oh, ow])
        if os.environ.get("AA_FUZZ_LOG"):  # soak runs: the case about to run, so that a crash names it
            with open(os.environ["AA_FUZZ_LOG"], "w") as lf:
                lf.write(repr((it, str(dt), c, (n, h, w), (oh, ow), cl, filt, kind)) + "\n")
        try:
            _lib.set_fused(1)
            y1 = call()
            v = _lib.last_variant()
            if os.environ.get("AA_FUZZ_LOG"):
                with open(os.environ["AA_FUZZ_LOG"], "a") as lf:
                    lf.write(v + " launched\n")
                torch.cuda.synchronize()
            _lib.set_fused(0)
            y0 = call()
        finally:
            _lib.set_fused(1)
        fused += v.startswith("fused")
        same = torch.equal(y1, y0) if y1.dtype not in (torch.float16, torch.bfloat16) else torch.equal(y1.view(torch.int16), y0.view(torch.int16))
        if not same and y1.is_floating_point():  # NaN-free inputs: any difference is a bug
            d = (y1.double() - y0.double()).abs().max().item()
            raise AssertionError((it, str(dt), c, (n, h, w), (oh, ow), cl, filt, kind, v, d))
        assert same, (it, str(dt), c, (n, h, w), (oh, ow), cl, filt, kind, v)
    assert fused >= 45 * cases // 120, fused  # about half of the random problems take a fused kernel (the rest: C = 2 or 5, fp64 channels_last, ...)


def test_fuzz_strong_downscales(aa):
    """80 seeded random STRONG down-scales of uint8 images in Pillow arithmetic (widths shrinking 7 .. 45 times: 17 .. 180 taps): the
    wide-window, split-window (four lanes per pixel, DPP reduction) and generic forms must all give the two-launch path's bytes; 1, 3, 4
    channels, both layouts, ragged sizes, windows clipped at both borders."""
    from interpolate_antialiasing_amd import _lib

    rng = np.random.default_rng(31415)
    seen = set()
    for it in range(80):
        c = int(rng.choice([1, 3, 3, 4]))
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(40, 500)), int(rng.integers(200, 2000))
        ow = max(1, int(w / rng.uniform(7.0, 45.0)))
        oh = max(1, int(h / rng.uniform(1.0, 12.0)))
        filt = ["linear", "cubic"][int(rng.integers(2))]
        x = torch.randint(0, 256, (n, c, h, w), dtype=torch.uint8, device="cuda")
        if bool(rng.integers(2)):
            x = x.contiguous(memory_format=torch.channels_last)
        fn = _fn(aa, filt)
        try:
            _lib.set_fused(1)
            y1 = fn(x, [oh, ow])
            seen.add(_lib.last_variant())
            _lib.set_fused(0)
            y0 = fn(x, [oh, ow])
        finally:
            _lib.set_fused(1)
        assert torch.equal(y1, y0), (it, c, (n, h, w), (oh, ow), filt, _lib.last_variant())
    assert {"fused_u8_nhwc_pil_v3", "fused_u8_planar_pil_v3"} <= seen, seen


def test_fuzz_vs_oracle_small(aa):
    """80 seeded random small problems straight against the oracle (bit-exact): fp32 / fp64 forward in both layouts, uint8 in Pillow
    and harness arithmetic, including sizes of 1, up-scales and windows clipped at both borders."""
    import os

    rng = np.random.default_rng(int(os.environ.get("AA_FUZZ_ORACLE_SEED", "77")))   # (a soak: AA_FUZZ_ORACLE_CASES=3000 AA_FUZZ_ORACLE_SEED=<n>)
    for it in range(int(os.environ.get("AA_FUZZ_ORACLE_CASES", "80"))):
        c = int(rng.choice([1, 3, 4]))
        n = int(rng.integers(1, 3))
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        oh = int(rng.integers(1, 120))
        ow = int(rng.integers(1, 120))
        filt = FILTS[int(rng.integers(3))]
        cl = bool(rng.integers(2))
        kind = int(rng.integers(4))
        if kind == 0:
            x = (rng.random((n, c, h, w)) * 300 - 40).astype(np.float32)
            exp = oracle.forward(filt, x, (oh, ow))
            got = _fn(aa, filt)(_gpu(x, cl), [oh, ow]).cpu().numpy()
        elif kind == 1:
            x = (rng.random((n, c, h, w)) * 300 - 40).astype(np.float64)
            exp = oracle.forward(filt, x, (oh, ow))
            got = _fn(aa, filt)(_gpu(x, cl), [oh, ow]).cpu().numpy()
        elif kind == 2:
            x = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
            exp = oracle.pil_resize_u8(filt, x, (oh, ow))
            got = _fn(aa, filt)(_gpu(x, cl), [oh, ow]).cpu().numpy()
        else:
            x = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
            exp = oracle.harness_u8(filt, x, (oh, ow))
            got = _fn(aa, filt)(_gpu(x, cl), [oh, ow], uint8_mode="harness").cpu().numpy()
        assert np.array_equal(got, exp), (it, kind, filt, (n, c, h, w), (oh, ow), cl)


# ------------------------------------------------------------------------------------------------ generic path, wide windows
def test_generic_wide_windows_vs_oracle(aa):
    """Strong down-scaling (17 .. 120 taps) lands in the generic two-launch path, whose horizontal pass then reads the window four
    elements per load (1, 3 or 4 interleaved channels) and whose vertical pass works a row per wave: bit-exact against the oracle
    for every dtype / arithmetic, both layouts, channel counts with and without the wide kernel, windows clipped at both borders
    (output sizes of 1 .. 3), and row counts that are not a multiple of the rows a thread walks at a time."""
    from interpolate_antialiasing_amd import _lib

    rng = np.random.default_rng(5)
    cases = [  # (C, channels_last, filter, (N, H, W), (oH, oW))
        (3, False, "cubic", (2, 41, 400), (9, 13)),
        (3, True, "linear", (2, 37, 500), (11, 17)),
        (4, True, "cubic", (1, 53, 301), (7, 10)),
        (1, False, "linear", (3, 33, 257), (3, 2)),
        (2, True, "cubic", (1, 30, 300), (5, 9)),      # 2 channels: the per-element kernel
        (5, True, "linear", (1, 19, 333), (4, 21)),
        (3, True, "box", (1, 45, 640), (6, 16)),
        (3, False, "linear", (1, 7, 1000), (1, 1)),
        (3, True, "cubic", (2, 300, 31), (10, 40)),     # wide windows in H only (W grows)
        (3, False, "linear", (1, 20, 300), (45, 1040)),  # growing heights, long rows: 16 uint8 / 4 floats per lane in the vertical pass
        (1, False, "cubic", (1, 9, 40), (30, 50)),       # short rows, fewer than 4 planes
        (3, False, "linear", (2, 12, 500), (33, 100)),   # short rows, planes walked 4 at a time (6 planes: a ragged last group)
        (3, True, "cubic", (1, 14, 900), (40, 120)),     # rows of 360 elements: 4 per lane, not 16
    ]
    generic = 0
    for c, cl, filt, (n, h, w), (oh, ow) in cases:
        f = _fn(aa, filt)
        xf = (rng.random((n, c, h, w)) * 300 - 40).astype(np.float32)
        xu = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
        got = f(_gpu(xf, cl), [oh, ow])
        generic += _lib.last_variant().startswith("generic")
        assert np.array_equal(got.cpu().numpy(), oracle.forward(filt, xf, (oh, ow))), ("f32", c, cl, filt, (n, h, w), (oh, ow))
        xd = xf.astype(np.float64) * 1.000001
        assert np.array_equal(f(_gpu(xd, cl), [oh, ow]).cpu().numpy(), oracle.forward(filt, xd, (oh, ow))), ("f64", c, cl, filt)
        got = f(_gpu(xu, cl), [oh, ow])
        generic += _lib.last_variant().startswith("generic")
        assert np.array_equal(got.cpu().numpy(), oracle.pil_resize_u8(filt, xu, (oh, ow))), ("pil", c, cl, filt, (n, h, w), (oh, ow))
        got = f(_gpu(xu, cl), [oh, ow], uint8_mode="harness")
        assert np.array_equal(got.cpu().numpy(), oracle.harness_u8(filt, xu, (oh, ow))), ("harness", c, cl, filt)
        for dt in (torch.float16, torch.bfloat16):  # 16-bit floats: fp32 arithmetic on the widened input, one rounding at the end
            xh = torch.from_numpy(xf).to(dt)
            exp = torch.from_numpy(oracle.forward(filt, xh.float().numpy(), (oh, ow))).to(dt)
            xg = xh.cuda().contiguous(memory_format=torch.channels_last) if cl else xh.cuda()
            assert torch.equal(f(xg, [oh, ow]).cpu(), exp), (str(dt), c, cl, filt)
    assert generic >= 12, generic  # (most of these shapes have no fused kernel: that is the point)


# ------------------------------------------------------------------------------------------------ uint8, heights that grow
def test_u8_growing_heights_take_the_fused_gather_form(aa):
    """Heights that grow (test.py's (120, 1200) and up-scaling in general): the fused uint8 kernel's vertical pass gathers over a
    register ring of the last input rows.  Bit-exact against the oracle in Pillow and harness arithmetic, channels_last (3 / 4
    channels) and planar, widths that grow or shrink, ragged output widths (byte stores), rows bands (many output rows), a
    single input row, and the float32-output conversion."""
    from interpolate_antialiasing_amd import _lib

    rng = np.random.default_rng(11)
    cases = [  # (C, channels_last, filter, (N, H, W), (oH, oW))
        (3, True, "linear", (2, 23, 50), (61, 128)),
        (3, True, "cubic", (2, 23, 50), (61, 128)),
        (3, True, "box", (1, 17, 40), (50, 64)),
        (4, True, "linear", (1, 19, 33), (47, 100)),
        (4, True, "cubic", (1, 30, 200), (77, 64)),     # W shrinks (13 taps), H grows
        (3, False, "linear", (1, 21, 64), (64, 200)),   # planar
        (3, False, "linear", (1, 21, 64), (64, 300)),   # planar, more than 256 columns: the generic path by choice
        (3, False, "cubic", (2, 9, 31), (40, 77)),      # planar, ragged width: byte stores
        (3, True, "linear", (1, 12, 70), (100, 45)),    # ragged width, channels_last
        (3, True, "linear", (1, 1, 16), (9, 32)),       # one input row
        (3, True, "cubic", (1, 40, 120), (700, 132)),   # many output rows: several bands
        (1, False, "linear", (3, 25, 906), (70, 120)),  # the harness's (120, H-up) shape class: 16-17 taps in W
    ]
    fused = 0
    for c, cl, filt, (n, h, w), (oh, ow) in cases:
        f = _fn(aa, filt)
        x = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
        got = f(_gpu(x, cl), [oh, ow])
        v = _lib.last_variant()
        fused += v.endswith("_v3")
        assert np.array_equal(got.cpu().numpy(), oracle.pil_resize_u8(filt, x, (oh, ow))), ("pil", v, c, cl, filt, (n, h, w), (oh, ow))
        got = f(_gpu(x, cl), [oh, ow], uint8_mode="harness")
        v = _lib.last_variant()
        fused += v.endswith("_v3")
        assert np.array_equal(got.cpu().numpy(), oracle.harness_u8(filt, x, (oh, ow))), ("harness", v, c, cl, filt, (n, h, w), (oh, ow))
        if c in (3, 4) and cl:  # decode-adjacent conversion: uint8 HWC in, float32 NCHW out == the oracle's fp32 forward
            got = f(_gpu(x, cl), [oh, ow], out_dtype=torch.float32, out_format="nchw")
            exp = oracle.forward(filt, x.astype(np.float32), (oh, ow))
            assert np.array_equal(got.cpu().numpy(), exp), ("to_f32", _lib.last_variant(), c, filt)
    assert fused >= 18, fused
    # at BASELINE-like size: fused == generic (two independent implementations), Pillow arithmetic, batch of 8
    x = torch.randint(0, 256, (8, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    # (harness arithmetic: <= 12 taps in W; Pillow arithmetic at 906 -> 120 belongs to the first-generation kernel: 200 columns here)
    for mode, sizes in (("pil", ([1200, 1200], [1200, 200])), ("harness", ([1200, 1200], [1200, 200]))):
        for size in sizes:
            try:
                _lib.set_fused(1)
                a = aa.linear_forward(x, size, uint8_mode=mode)
                va = _lib.last_variant()
                _lib.set_fused(0)
                b = aa.linear_forward(x, size, uint8_mode=mode)
                vb = _lib.last_variant()
            finally:
                _lib.set_fused(1)
            assert va.endswith("_v3") and vb.startswith("generic"), (va, vb)
            assert torch.equal(a, b), (size, mode, va, vb)


def test_backward_store_forms_agree(aa):
    """The up-scaling / backward kernel stores large outputs with the streaming policy in three forms chosen from the row pitch:
    whole pieces (rows of whole 64-byte sectors), pieces cut at each row's sector boundaries through an LDS shift (even widths),
    and split policies with a pacing barrier (odd widths).  aa_set_store_form(1) applies them at test sizes: each must equal the generic
    path bit for bit, for widths that end strips in every way (exactly full, two columns over, a short last strip)."""
    from interpolate_antialiasing_amd import _lib

    torch.manual_seed(9)
    assert _lib.set_store_form(1) == -1
    try:
        for (h, w), (oh, ow) in (((196, 320), (438, 906)), ((50, 100), (131, 466)), ((50, 100), (77, 468)), ((40, 200), (90, 494)),
                                 ((60, 90), (61, 258)), ((33, 300), (100, 905)), ((20, 128), (64, 512)), ((30, 400), (65, 1202))):
            g = torch.randn(5, 3, h, w, device="cuda")
            for fn in (aa.linear_backward, aa.cubic_backward):
                _lib.set_fused(1)
                a = fn(g, [h, w], [5, 3, oh, ow])
                va = _lib.last_variant()
                _lib.set_fused(0)
                b = fn(g, [h, w], [5, 3, oh, ow])
                _lib.set_fused(1)
                assert va == "fused_f32_nchw_up", (va, (h, w), (oh, ow))
                assert torch.equal(a, b), ((h, w), (oh, ow), fn.__name__)
            x = torch.rand(3, 2, h, w, device="cuda") * 255 - 20  # forward up-scaling through the same kernel
            _lib.set_fused(1)
            a = aa.linear_forward(x, [oh, ow])
            va = _lib.last_variant()
            _lib.set_fused(0)
            b = aa.linear_forward(x, [oh, ow])
            _lib.set_fused(1)
            assert va == "fused_f32_nchw_up" and torch.equal(a, b), (va, (h, w), (oh, ow))
    finally:
        _lib.set_fused(1)
        assert _lib.set_store_form(-1) == 1


def test_fast_precision_mode_is_within_tolerance(aa, golden_forward):
    """precision="fast" (AA_FLAG_FAST): FMA accumulation over zero-padded windows.  Bar: BASELINE.json's float tolerance, 1e-4
    relative to the reference's CPU path (here: the oracle, which the exact mode equals bit for bit) on BASELINE configs 0 and 2 and
    the reference-derived goldens; a fast kernel must really have run for the plane layouts; the default stays bit-exact.
    Documented non-finite behaviour: a NaN reaches every output whose window holds it (as in exact mode) and may additionally reach
    outputs whose 16-byte-aligned read window holds it — never anything further than 3 columns beyond, never another row band."""
    from interpolate_antialiasing_amd import _lib

    rng = np.random.default_rng(77)
    for filt, shape, size in (("linear", (2, 3, 438, 906), (196, 320)), ("cubic", (1, 3, 1024, 1024), (224, 224)),
                              ("cubic", (2, 3, 61, 53), (17, 23)), ("linear", (2, 3, 100, 300), (37, 128))):
        x = (rng.random(shape, dtype=np.float32) * 255).astype(np.float32)
        exp = oracle.forward(filt, x, size, nthreads=8)
        xt = _gpu(x)
        y_exact = _fn(aa, filt)(xt, list(size))
        assert np.array_equal(y_exact.cpu().numpy(), exp), (filt, shape)
        y_fast = _fn(aa, filt)(xt, list(size), precision="fast")
        assert _lib.last_variant() == "fused_f32_nchw_fast", (_lib.last_variant(), filt, shape)
        np.testing.assert_allclose(y_fast.cpu().numpy(), exp, rtol=1e-4, atol=1e-4 * 255)  # tolerance of the north star, stated here
        rel = np.abs(y_fast.cpu().numpy() - exp).max() / 255.0
        assert rel < 1e-5, rel  # (what it actually is: rounding only)
        # 16-bit floats: half(fast_fp32) against half(oracle): one unit in the last place at most
        xh = xt.half()
        yh = _fn(aa, filt)(xh, list(size), precision="fast")
        if filt == "linear":  # (21-tap bicubic windows are beyond the 16-bit kernels' 17 taps: generic path, exact arithmetic)
            assert _lib.last_variant() == "fused_f16_nchw_fast", _lib.last_variant()
        yh_exact = _fn(aa, filt)(xh, list(size))
        np.testing.assert_allclose(yh.float().cpu().numpy(), yh_exact.float().cpu().numpy(), rtol=2e-3, atol=0.25)
    # the reference-derived goldens (outputs of the reference's own build), fp32 plane layout
    n = 0
    for case in sorted({k.split("_")[0] for k in golden_forward.files}):
        x = golden_forward[f"{case}_x"]
        size = [int(v) for v in golden_forward[f"{case}_size"]]
        ac = bool(golden_forward[f"{case}_align"])
        for filt in FILTS:
            exp = golden_forward[f"{case}_{filt}_f32"]
            got = _fn(aa, filt)(_gpu(x), size, ac, precision="fast").cpu().numpy()
            np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(exp).max())))
            n += 1
    assert n >= 15
    # layouts / dtypes without a tolerance kernel run the exact ones (always within tolerance), uint8 ignores the flag
    xcl = _gpu((rng.random((2, 3, 120, 200), dtype=np.float32) * 255).astype(np.float32), channels_last=True)
    assert torch.equal(aa.linear_forward(xcl, [50, 80], precision="fast"), aa.linear_forward(xcl, [50, 80]))
    xd = torch.rand(1, 2, 90, 130, device="cuda", dtype=torch.float64)
    assert torch.equal(aa.linear_forward(xd, [40, 60], precision="fast"), aa.linear_forward(xd, [40, 60]))
    x8 = torch.randint(0, 256, (2, 3, 120, 200), dtype=torch.uint8, device="cuda")
    assert torch.equal(aa.linear_forward(x8, [50, 80], precision="fast"), aa.linear_forward(x8, [50, 80]))
    with pytest.raises(ValueError):
        aa.linear_forward(xcl, [50, 80], precision="sloppy")
    # uint8 in float arithmetic (the harness's semantics, the decode-adjacent conversion): the tolerance kernels really run; the float32
    # output is within 1e-4 relative of the exact one, the truncated byte() within one count (and mostly identical)
    x8 = torch.randint(0, 256, (3, 438, 906, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)
    for layout in ("nhwc", "nchw"):
        xin = x8 if layout == "nhwc" else x8.contiguous()
        f_exact = aa.linear_forward(xin, [196, 320], out_dtype=torch.float32, out_format="nchw")
        f_fast = aa.linear_forward(xin, [196, 320], out_dtype=torch.float32, out_format="nchw", precision="fast")
        assert _lib.last_variant().endswith("_v3_fast"), _lib.last_variant()
        np.testing.assert_allclose(f_fast.cpu().numpy(), f_exact.cpu().numpy(), rtol=1e-4, atol=1e-4 * 255)
        b_exact = aa.linear_forward(xin, [196, 320], uint8_mode="harness")
        b_fast = aa.linear_forward(xin, [196, 320], uint8_mode="harness", precision="fast")
        assert _lib.last_variant().endswith("harness_v3_fast"), _lib.last_variant()
        diff = (b_fast.int() - b_exact.int()).abs()
        assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 1e-3
    # non-finite behaviour
    x = (rng.random((1, 1, 120, 400), dtype=np.float32) * 255).astype(np.float32)
    x[0, 0, 60, 200] = np.nan
    xt = _gpu(x)
    ne = torch.isnan(aa.linear_forward(xt, [50, 100])).cpu().numpy()[0, 0]
    nf = torch.isnan(aa.linear_forward(xt, [50, 100], precision="fast")).cpu().numpy()[0, 0]
    assert ne.any() and (nf | ~ne).all(), "fast mode must poison at least the outputs exact mode poisons"
    rows_e, cols_e = np.nonzero(ne.any(1))[0], np.nonzero(ne.any(0))[0]
    rows_f, cols_f = np.nonzero(nf.any(1))[0], np.nonzero(nf.any(0))[0]
    assert rows_f.min() == rows_e.min() and rows_f.max() == rows_e.max()  # vertical windows are exact in both modes
    assert cols_f.min() >= cols_e.min() - 3 and cols_f.max() <= cols_e.max() + 3, (cols_e, cols_f)


def test_wide_windows_take_the_fused_uint8_kernel(aa, golden_kat):
    """test.py's own thumbnail sizes (test.py:15-21: (120, 96) -> 17 bilinear / 33 bicubic taps along W) and other strong down-scales:
    uint8 in Pillow arithmetic now runs the fused single-launch kernel with a 24- or 34-tap window (up to 6 open output rows), for
    channels_last and planar bytes: bit-identical to Pillow (oracle) and to the generic two-launch path."""
    from interpolate_antialiasing_amd import _lib

    rgb = golden_kat["rgb"]  # the reference's test.png, 438 x 906 x 3
    x_hwc = torch.from_numpy(np.ascontiguousarray(rgb)).cuda()[None].repeat(3, 1, 1, 1)
    x_hwc[1] = torch.flip(x_hwc[1], dims=(1,))
    x_hwc[2] = torch.randint(0, 256, x_hwc[2].shape, dtype=torch.uint8, device="cuda")
    for filt in ("linear", "cubic"):
        for size in ((96, 120), (60, 200), (30, 115), (200, 130)):
            for planar in (False, True):
                x = x_hwc.permute(0, 3, 1, 2)
                x = x.contiguous() if planar else x
                try:
                    _lib.set_fused(1)
                    y1 = _fn(aa, filt)(x, list(size))
                    v1 = _lib.last_variant()
                    _lib.set_fused(0)
                    y0 = _fn(aa, filt)(x, list(size))
                finally:
                    _lib.set_fused(1)
                assert v1 == ("fused_u8_planar_pil_v3" if planar else "fused_u8_nhwc_pil_v3"), (v1, filt, size, planar)
                assert torch.equal(y1, y0), (filt, size, planar)
                exp = oracle.pil_resize_u8(filt, x.cpu().numpy(), size)
                assert np.array_equal(y1.cpu().numpy(), exp), (filt, size, planar)
    # the same windows in the harness's float arithmetic (uint8 out) and as float32 planes out (round 3): bit-identical to the two-launch
    # path and to the oracle's harness restatement; precision="fast" runs the exact wide kernel (no tolerance instantiation that wide)
    for filt in ("linear", "cubic"):
        for size in ((96, 120), (60, 200)):
            for planar in (False, True):
                x = x_hwc.permute(0, 3, 1, 2)
                x = x.contiguous() if planar else x
                for kw, want in (({"uint8_mode": "harness"}, "fused_u8_planar_harness_v3" if planar else "fused_u8_nhwc_harness_v3"),
                                 ({"out_dtype": torch.float32, "out_format": "nchw"}, "fused_u8_planar_to_f32_v3" if planar else "fused_u8_nhwc_to_f32_nchw_v3")):
                    try:
                        _lib.set_fused(1)
                        y1 = _fn(aa, filt)(x, list(size), **kw)
                        v1 = _lib.last_variant()
                        yf = _fn(aa, filt)(x, list(size), precision="fast", **kw)
                        vf = _lib.last_variant()
                        _lib.set_fused(0)
                        y0 = _fn(aa, filt)(x, list(size), **kw)
                    finally:
                        _lib.set_fused(1)
                    assert v1 == want and vf in (want, want + "_fast"), (v1, vf, want, filt, size)  # (_fast: windows of up to 16 taps)
                    assert torch.equal(y1, y0), (filt, size, planar, want)
                    if vf == want:
                        assert torch.equal(yf, y0), (filt, size, planar, want)
                    elif yf.dtype == torch.uint8:
                        assert (yf.int() - y0.int()).abs().max().item() <= 1
                    else:
                        torch.testing.assert_close(yf, y0, rtol=1e-4, atol=1e-3)
                    if "uint8_mode" in kw:
                        assert np.array_equal(y1.cpu().numpy(), oracle.harness_u8(filt, x.cpu().numpy(), size)), (filt, size, planar)
    # fp32 planes: 33-tap windows (9 aligned reads per row, lane masks in vector registers) on strips of 32 columns
    xf = x_hwc.permute(0, 3, 1, 2).float().contiguous()
    for filt, size in (("cubic", (96, 120)), ("cubic", (200, 130)), ("linear", (60, 110)), ("linear", (60, 52)), ("cubic", (90, 100))):  # (the last two: 35-41 taps, 11 reads, strips of 16 / 32 columns)
        y = _fn(aa, filt)(xf, list(size))
        assert _lib.last_variant() == "fused_f32_nchw", (_lib.last_variant(), filt, size)
        assert np.array_equal(y.cpu().numpy(), oracle.forward(filt, xf.cpu().numpy(), size, nthreads=8)), (filt, size)
        yf = _fn(aa, filt)(xf, list(size), precision="fast")
        assert _lib.last_variant() == "fused_f32_nchw_fast"
        np.testing.assert_allclose(yf.cpu().numpy(), y.cpu().numpy(), rtol=1e-4, atol=1e-4 * 255)
    # beyond 34 taps: SPLIT windows (round 3) — four lanes share an output pixel, each holds a quarter of its window, the partial sums meet
    # in two DPP additions (BASELINE north_star's "wavefront shuffles to reduce the variable-width filter tap accumulation"): integer sums
    # are associative, so Pillow's result bit for bit.  35 .. 136 taps, 16 / 24 / 34 taps per lane, channels_last and planar.
    for filt, size in (("cubic", (40, 40)), ("linear", (20, 30)), ("cubic", (33, 61)), ("linear", (12, 30)), ("cubic", (100, 28)), ("linear", (196, 26))):
        for planar in (False, True):
            x = x_hwc.permute(0, 3, 1, 2)
            x = x.contiguous() if planar else x
            try:
                _lib.set_fused(1)
                y1 = _fn(aa, filt)(x, list(size))
                v1 = _lib.last_variant()
                _lib.set_fused(0)
                y0 = _fn(aa, filt)(x, list(size))
            finally:
                _lib.set_fused(1)
            assert v1 == ("fused_u8_planar_pil_v3" if planar else "fused_u8_nhwc_pil_v3"), (v1, filt, size, planar)
            assert torch.equal(y1, y0), (filt, size, planar)
            assert np.array_equal(y1.cpu().numpy(), oracle.pil_resize_u8(filt, x.cpu().numpy(), size)), (filt, size, planar)
    # beyond 136 taps: the generic path
    y = aa.cubic_forward(x_hwc.permute(0, 3, 1, 2), [40, 12])
    assert _lib.last_variant().startswith("generic"), _lib.last_variant()
    assert np.array_equal(y.cpu().numpy(), oracle.pil_resize_u8("cubic", x_hwc.permute(0, 3, 1, 2).cpu().numpy(), (40, 12)))


def test_strided_views_are_read_in_place(aa, monkeypatch):
    """The reference walks arbitrary strides through TensorIterator (s2.2/aa_interpolation_impl.h:555-559).  The views a data pipeline
    produces — a crop of a larger tensor (RandomResizedCrop), a batch slice — go through aa_resample_fwd_strided: the fused kernels read
    them where they lie (rows dense, any row / image pitch), with no .contiguous() round trip; results equal those of the dense copy
    bit for bit.  Views no kernel takes (planes not uniformly spaced, no fused kernel for the shape) still work, through the copy."""
    from interpolate_antialiasing_amd import _lib

    copies = []
    real = aa._memory_format
    monkeypatch.setattr(aa, "_memory_format", lambda t: (copies.append(tuple(t.shape)), real(t))[1])
    torch.manual_seed(13)
    big8 = torch.randint(0, 256, (5, 500, 1000, 3), dtype=torch.uint8, device="cuda").permute(0, 3, 1, 2)  # channels_last storage
    bigf = torch.rand(3, 2, 500, 1000, device="cuda") * 255
    cases = [
        (big8[:, :, 31:469, 47:953], [196, 320], {}, "fused_u8_nhwc_pil_v3"),                       # crop, odd byte offsets
        (big8[1::2, :, 31:469, 47:953], [196, 320], {}, "fused_u8_nhwc_pil_v3"),                    # crop of a batch slice
        (big8[:, :, 2:440, 1:907], [196, 320], {"uint8_mode": "harness"}, "fused_u8_nhwc_harness_v3"),
        (big8[:, :, 0:438, 0:906], [196, 320], {"out_dtype": None}, "fused_u8_nhwc_pil_v3"),        # crop at the origin: only the pitch differs
        (bigf[:, :, 10:448, 20:926], [196, 320], {}, "fused_f32_nchw"),                             # fp32 planes
        (bigf[:, :, 10:448, 21:927], [196, 320], {"precision": "fast"}, "fused_f32_nchw_fast"),
        (bigf.half()[:, :, 5:443, 3:909], [196, 320], {}, "fused_f16_nchw"),                        # rows starting on odd halves
        (bigf[1:2, :, 10:448, 20:926], [120, 200], {}, "fused_f32_nchw"),
    ]
    for view, size, kw, want in cases:
        kw = {k: v for k, v in kw.items() if v is not None}
        assert not view.is_contiguous() and not view.is_contiguous(memory_format=torch.channels_last)
        copies.clear()
        y = aa.linear_forward(view, size, **kw)
        assert _lib.last_variant() == want, (_lib.last_variant(), want, tuple(view.shape), view.stride())
        assert copies == [], "the view was copied"
        dense = view.contiguous(memory_format=torch.channels_last) if view.stride(1) == 1 else view.contiguous()
        ref = aa.linear_forward(dense, size, **kw)
        assert torch.equal(y, ref), (want, tuple(view.shape))
        assert y.is_contiguous(memory_format=torch.channels_last) == ref.is_contiguous(memory_format=torch.channels_last)
    x = big8[:, :, 31:469, 47:953]
    exp = oracle.pil_resize_u8("linear", x[:1].cpu().numpy(), (196, 320))
    assert np.array_equal(aa.linear_forward(x, [196, 320])[:1].cpu().numpy(), exp)
    # views that are copied: a batch slice of planes (not uniformly spaced), a transposed view, fp64 growing heights (no fused kernel)
    for view, size in ((bigf[::2, :, 10:448, 20:926], [196, 320]), (bigf.transpose(2, 3)[:, :, 20:926, 10:448], [320, 196]),
                       (bigf.double()[:, :, 10:100, 20:200], [200, 100])):
        copies.clear()
        y = aa.linear_forward(view, size)
        assert len(copies) == 1
        assert torch.equal(y, aa.linear_forward(view.contiguous(), size))


def test_tensors_beyond_4_gib(aa):
    """The reference indexes with int64_t throughout (s2.2/aa_interpolation_impl.h:688-699); SURVEY 8(d) config 4 in fp32 is 4.88 GB
    per GPU.  The kernels address an image / plane with 32-bit offsets from a 64-bit base and clamp their buffer ranges: here a
    tensor's bytes run past 4 GiB (and past 2^31 elements for uint8), for both headline layouts.  Every image: fused == generic
    two-launch path (independent implementations); first / middle / last image and the first one beyond the 4 GiB mark: == oracle."""
    from interpolate_antialiasing_amd import _lib

    free, _ = torch.cuda.mem_get_info()
    if free < 24 * (1 << 30):
        pytest.skip(f"needs ~24 GiB of free HBM, {free >> 30} GiB free")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    # fp32 NCHW [1024,3,906,438] -> [320,196] (4.88 GB in), uint8 channels_last [4096,3,438,906] -> [196,320] (4.88 GB, 4.9e9 elements)
    for case in ("f32", "u8"):
        if case == "f32":
            x = torch.rand((1024, 3, 906, 438), device="cuda", generator=gen) * 255
            size, img_bytes = [320, 196], 3 * 906 * 438 * 4
        else:
            x = torch.randint(0, 256, (4096, 438, 906, 3), dtype=torch.uint8, device="cuda", generator=gen).permute(0, 3, 1, 2)
            size, img_bytes = [196, 320], 3 * 438 * 906
        assert x.numel() * x.element_size() > (1 << 32)
        try:
            _lib.set_fused(1)
            y1 = aa.linear_forward(x, size)
            v1 = _lib.last_variant()
            _lib.set_fused(0)
            y0 = aa.linear_forward(x, size)
            v0 = _lib.last_variant()
        finally:
            _lib.set_fused(1)
        assert v1.startswith("fused") and v0.startswith("generic"), (v1, v0)
        assert torch.equal(y1, y0), (case, v1, v0)
        del y0
        n = x.shape[0]
        beyond = (1 << 32) // img_bytes + 1  # the first image that lies entirely past the 4 GiB mark
        for i in sorted({0, n // 2, beyond, n - 1}):
            xi = x[i:i + 1].cpu().numpy()
            exp = oracle.forward("linear", xi, tuple(size), nthreads=8) if case == "f32" else oracle.pil_resize_u8("linear", xi, tuple(size), nthreads=8)
            assert np.array_equal(y1[i:i + 1].cpu().numpy(), exp), (case, i)
        if case == "f32":  # the true adjoint at the same size: 0.77 GB of gradients -> 4.88 GB
            g = torch.randn((1024, 3, 320, 196), device="cuda", generator=gen)
            gi = aa.linear_backward(g, size, [1024, 3, 906, 438])
            for i in (0, beyond, 1023):
                exp = oracle.backward("linear", g[i:i + 1].cpu().numpy(), (906, 438))
                assert np.abs(gi[i:i + 1].cpu().numpy() - exp).max() < 1e-4, i
            del g, gi
        del x, y1
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_bench_rank_path_through_the_launcher():
    """The multi-rank code path end to end on hardware, as a fresh process: `bench.py --gpus 1 --force-launcher` makes the parent
    (which never touches the GPU) start one rank that joins a 1-rank RCCL process group, receives the tables by broadcast, runs the
    timed loop between barriers and reduces its timing — everything an N-GPU run does except a second rank."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-launcher", "--batch", "8", "--steps", "3",
                          "--warmup", "1", "--prewarm-seconds", "0", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["process_group"] == "nccl" and line["max_abs_err_vs_oracle"] == 0
    assert line["value"] > 0 and line["secondary"] is None and line["cpu_baseline"] is None
    assert line["roofline"]["read_frac"] < line["roofline"]["frac"]

// extension_interpolate_amd.cpp — the pybind11 module a maintainer of the reference would build INSTEAD of
// step_two_dot_two/extension_interpolate.cpp (same four callables, same signatures: :7-42, :46-51), with every body replaced
// by calls into libaa_interp.so's C-ABI (include/aa_interp.h).  INTEGRATION.md section 2 quotes this file; it is compiled by
// __graft_entry__.build() (tools/integration_stub/build.py) and exercised on the GPU by
// tests/test_gpu_parity.py::test_integration_stub_matches_the_shim, so the documentation cannot rot.
//
// Written for this repository (it is not part of the reference).  Tables are cached per (filter, kind, in, out,
// align_corners, device): the reference rebuilds its weights on every call and every pass (aa_interpolation_impl.h:195-281).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include <map>
#include <mutex>
#include <tuple>

#include "aa_interp.h"

namespace {

struct Table {
  torch::Tensor buf;  // packed table in HBM (keeps the memory alive)
  aa_axis axis;
};

const Table &get_table(int filter, int kind, int64_t in, int64_t out, bool align_corners, const torch::Device &dev,
                       hipStream_t stream, bool transposed = false) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int64_t, int64_t, bool, int, bool>, Table> cache;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(filter, kind, in, out, align_corners, (int)dev.index(), transposed);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;

  const int k = aa_table_ksize(filter, kind, in, out, align_corners, 0.0);
  TORCH_CHECK(k > 0, aa_strerror(k));
  const size_t nbytes = aa_table_build_bytes(filter, kind, in, out, align_corners, 0.0);
  auto fwd = torch::empty({(int64_t)nbytes}, torch::dtype(torch::kUInt8).device(dev));
  int rc = aa_table_build(filter, kind, in, out, align_corners, 0.0, fwd.data_ptr(), nbytes, stream);
  TORCH_CHECK(rc == AA_OK, aa_strerror(rc));
  Table t;
  aa_table_header h;
  if (!transposed) {
    rc = aa_table_query(fwd.data_ptr(), &h, stream);  // one 64-byte read-back per (shape, filter), then cached
    TORCH_CHECK(rc == AA_OK, aa_strerror(rc));
    t.buf = fwd;
  } else {  // the adjoint (gather-form) table the backward uses: in/out swapped
    const int tk = aa_table_transposed_ksize(filter, kind, in, out, align_corners, 0.0);
    TORCH_CHECK(tk > 0, aa_strerror(tk));
    const size_t tbytes = aa_table_bytes(kind, in, tk);
    t.buf = torch::empty({(int64_t)tbytes}, torch::dtype(torch::kUInt8).device(dev));
    rc = aa_table_transpose(fwd.data_ptr(), t.buf.data_ptr(), tbytes, tk, stream);
    TORCH_CHECK(rc == AA_OK, aa_strerror(rc));
    rc = aa_table_query(t.buf.data_ptr(), &h, stream);
    TORCH_CHECK(rc == AA_OK, aa_strerror(rc));
  }
  t.axis = aa_axis{t.buf.data_ptr(), h.in_size,  h.out_size,      h.ksize,       h.max_taps, h.kind,
                   h.filter,         h.scatter_off, h.scatter_ksize, h.scatter_max, h.span64p1, h.span4p1, h.gather_off, {0, 0}};
  return cache.emplace(key, std::move(t)).first->second;
}

int dtype_of(const torch::Tensor &x) {
  switch (x.scalar_type()) {
    case at::kByte: return AA_U8;
    case at::kFloat: return AA_F32;
    case at::kDouble: return AA_F64;
    case at::kHalf: return AA_F16;
    case at::kBFloat16: return AA_BF16;
    default: TORCH_CHECK_NOT_IMPLEMENTED(false, "\"upsample_generic_Nd\" not implemented for '", toString(x.scalar_type()), "'");
  }
}

torch::Tensor forward_impl(int filter, const torch::Tensor &input, at::IntArrayRef output_size, bool align_corners) {
  TORCH_CHECK(output_size.size() == 2, "It is expected output_size equals to 2, but got size ", output_size.size());
  TORCH_CHECK(input.dim() == 4, "It is expected input_size equals to 4, but got size ", input.dim());
  TORCH_CHECK(input.is_cuda(), "the MI355X path takes tensors on a ROCm GPU");
  const bool cl = !input.is_contiguous() && input.is_contiguous(at::MemoryFormat::ChannelsLast);
  const auto x = cl ? input : input.contiguous();
  const int dtype = dtype_of(x);
  // uint8: Pillow-exact integer arithmetic; floats: the reference's arithmetic in the tensor's own width (halves in fp32)
  const int kind = dtype == AA_U8 ? AA_TABLE_PIL : (dtype == AA_F64 ? AA_TABLE_F64 : AA_TABLE_F32);
  const int64_t N = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3), oH = output_size[0], oW = output_size[1];
  TORCH_CHECK(H > 0 && W > 0 && oH > 0 && oW > 0, "Input and output sizes should be greater than 0, but got input (H: ", H,
              ", W: ", W, ") output (H: ", oH, ", W: ", oW, ")");
  const c10::hip::HIPStream cur = c10::hip::getCurrentHIPStream(x.device().index());
  hipStream_t stream = cur.stream();
  auto out = torch::empty({N, C, oH, oW}, x.options().memory_format(cl ? at::MemoryFormat::ChannelsLast : at::MemoryFormat::Contiguous));
  if (N == 0) return out;
  const Table &th = get_table(filter, kind, H, oH, align_corners, x.device(), stream);
  const Table &tw = get_table(filter, kind, W, oW, align_corners, x.device(), stream);
  const int layout = cl ? AA_NHWC : AA_NCHW;
  const size_t ws_bytes = aa_workspace_bytes(dtype, layout, N, C, H, W, oH, oW, &th.axis, &tw.axis);
  auto ws = torch::empty({(int64_t)ws_bytes}, torch::dtype(torch::kUInt8).device(x.device()));
  const int rc = aa_resample_fwd(x.data_ptr(), out.data_ptr(), ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, dtype, layout, N, C, H, W,
                                 &th.axis, &tw.axis, stream);
  TORCH_CHECK(rc == AA_OK, aa_strerror(rc));
  return out;
}

}  // namespace

// ---- the reference's four callables (step_two_dot_two/extension_interpolate.cpp:7-42) -----------------------------------
torch::Tensor interpolate_linear_forward(const torch::Tensor &input, at::IntArrayRef output_size, bool align_corners = false) {
  return forward_impl(AA_FILTER_LINEAR, input, output_size, align_corners);
}
torch::Tensor interpolate_nearest_forward(const torch::Tensor &input, at::IntArrayRef output_size, bool align_corners = false) {
  return forward_impl(AA_FILTER_BOX, input, output_size, align_corners);  // "it's not nearest but box" (:48)
}
torch::Tensor interpolate_cubic_forward(const torch::Tensor &input, at::IntArrayRef output_size, bool align_corners = false) {
  return forward_impl(AA_FILTER_CUBIC, input, output_size, align_corners);
}
torch::Tensor interpolate_linear_backward(const torch::Tensor &grad_output, at::IntArrayRef output_size, at::IntArrayRef input_size,
                                          bool align_corners = false) {
  TORCH_CHECK(output_size.size() == 2 && input_size.size() == 4 && grad_output.dim() == 4 && grad_output.is_cuda());
  const int64_t N = input_size[0], C = input_size[1], H = input_size[2], W = input_size[3], oH = output_size[0], oW = output_size[1];
  TORCH_CHECK(grad_output.size(0) == N && grad_output.size(1) == C && grad_output.size(2) == oH && grad_output.size(3) == oW,
              "Expected grad_output to have the same shape as output");
  const bool cl = !grad_output.is_contiguous() && grad_output.is_contiguous(at::MemoryFormat::ChannelsLast);
  const auto go = cl ? grad_output : grad_output.contiguous();
  const int dtype = dtype_of(go);
  TORCH_CHECK_NOT_IMPLEMENTED(dtype == AA_F32 || dtype == AA_F64, "backward takes float or double gradients");
  const int kind = dtype == AA_F64 ? AA_TABLE_F64 : AA_TABLE_F32;
  hipStream_t stream = c10::hip::getCurrentHIPStream(go.device().index()).stream();
  auto gi = torch::empty({N, C, H, W}, go.options().memory_format(cl ? at::MemoryFormat::ChannelsLast : at::MemoryFormat::Contiguous));
  if (N == 0) return gi;
  // the TRUE adjoint of the antialiased forward (the reference's header is the non-AA one): gather form through the
  // transposed tables, i.e. a forward resample of grad_output
  const Table &trh = get_table(AA_FILTER_LINEAR, kind, H, oH, align_corners, go.device(), stream, /*transposed=*/true);
  const Table &trw = get_table(AA_FILTER_LINEAR, kind, W, oW, align_corners, go.device(), stream, /*transposed=*/true);
  const int layout = cl ? AA_NHWC : AA_NCHW;
  const size_t ws_bytes = aa_workspace_bytes(dtype, layout, N, C, oH, oW, H, W, &trh.axis, &trw.axis);
  auto ws = torch::empty({(int64_t)ws_bytes}, torch::dtype(torch::kUInt8).device(go.device()));
  const int rc = aa_resample_bwd(go.data_ptr(), gi.data_ptr(), ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, dtype, layout, N, C, H, W,
                                 &trh.axis, &trw.axis, stream);
  TORCH_CHECK(rc == AA_OK, aa_strerror(rc));
  return gi;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("linear_forward", &interpolate_linear_forward, "Anti-Alias Linear Interpolation forward (MI355X)", py::arg("input"),
        py::arg("output_size"), py::arg("align_corners") = false);
  m.def("nearest_forward", &interpolate_nearest_forward, "Anti-Alias Nearest (box) Interpolation forward (MI355X)", py::arg("input"),
        py::arg("output_size"), py::arg("align_corners") = false);
  m.def("cubic_forward", &interpolate_cubic_forward, "Anti-Alias Cubic Interpolation forward (MI355X)", py::arg("input"),
        py::arg("output_size"), py::arg("align_corners") = false);
  m.def("linear_backward", &interpolate_linear_backward, "Anti-Alias Linear Interpolation backward (true adjoint, MI355X)",
        py::arg("grad_output"), py::arg("output_size"), py::arg("input_size"), py::arg("align_corners") = false);
}

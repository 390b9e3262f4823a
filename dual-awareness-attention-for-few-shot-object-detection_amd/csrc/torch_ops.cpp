// torch.ops.dana.* -- the reference's pybind module `model._C` (lib/model/csrc/vision.cpp:7-13: nms,
// roi_align_forward / backward, roi_pool_forward / backward) registered as PyTorch custom operators over the C ABI of
// libdana_hip.so (include/dana_hip.h). Host-only C++: no kernels here, every operator validates its tensors, allocates
// the outputs with ATen, takes the CURRENT HIP stream of the input's device and calls the matching dana_* entry point.
// RoIAlign / RoIPool also get autograd formulas (what lib/model/roi_layers/roi_align.py:12-43 / roi_pool.py build with
// torch.autograd.Function), so `torch.ops.dana.roi_align(input, rois, ...)` is differentiable w.r.t. `input`.
//
// (PyTorch's ROCm build keeps the device type "cuda": the guard / stream accessors are the *MasqueradingAsCUDA ones.)
//
// Errors follow the reference's contract: a failed dana_* call becomes a C++ exception -> Python
// RuntimeError carrying dana_last_error(); CPU tensors raise ("Not compiled with CPU support": this build has no CPU
// kernels); empty inputs give empty results without a launch (nms.h:17-18, ROIAlign_cuda.cu:278-281).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/dana_hip.h"

namespace {

void check(int rc, const char* who) {
  TORCH_CHECK(rc == 0, who, " failed (", rc, "): ", dana_last_error());
}

const at::Tensor& on_gpu(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a HIP tensor: Not compiled with CPU support (this build has no CPU kernels)");
  return t;
}

dana_stream_t stream_of(const at::Tensor& t) {
  return (dana_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

// nms.h:10-28: dets [N,4], scores [N] -> int64 kept ORIGINAL indices, ascending; IoU > threshold suppresses (nms.cu:60)
at::Tensor nms(const at::Tensor& dets, const at::Tensor& scores, double threshold) {
  if (dets.numel() == 0) return at::empty({0}, dets.options().dtype(at::kLong).device(at::kCPU));
  on_gpu(dets, "dets");
  on_gpu(scores, "scores");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(dets.device());
  const auto d = dets.contiguous().to(at::kFloat), s = scores.contiguous().to(at::kFloat);
  const int n = (int)d.size(0);
  TORCH_CHECK(d.dim() == 2 && d.size(1) == 4 && s.numel() == n, "nms: dets [N,4], scores [N]");
  auto order = at::empty({1, n}, d.options().dtype(at::kInt));
  auto sorted = at::empty({1, n}, d.options());
  auto ws = at::empty({(int64_t)dana_sort_desc_workspace_bytes(1, n) + 16}, d.options().dtype(at::kByte));
  check(dana_sort_desc(s.data_ptr<float>(), 1, n, order.data_ptr<int>(), sorted.data_ptr<float>(), ws.data_ptr(),
                       (size_t)ws.numel(), stream_of(d)), "dana_sort_desc");
  const auto idx = order.view({n}).to(at::kLong);
  const auto boxes = d.index_select(0, idx).contiguous();
  auto keep = at::empty({n}, d.options().dtype(at::kInt));
  auto num = at::empty({1}, d.options().dtype(at::kInt));
  auto ws2 = at::empty({(int64_t)dana_nms_workspace_bytes(n, 1) + 16}, d.options().dtype(at::kByte));
  check(dana_nms(boxes.data_ptr<float>(), n, 1, (float)threshold, 0, n, keep.data_ptr<int>(), n, num.data_ptr<int>(),
                 ws2.data_ptr(), (size_t)ws2.numel(), stream_of(d)), "dana_nms");
  const int k = num.item<int>();  // the variable-length result needs the count on the host, as the reference's host scan
  const auto kept = idx.index_select(0, keep.slice(0, 0, k).to(at::kLong));
  if (k == 0 || n >= (1 << 24)) return k == 0 ? kept : std::get<0>(at::sort(kept));
  // ascending ORIGINAL indices (nms.cu:125-131): this library's own sort on the negated indices (exact in fp32 below 2^24)
  const auto neg = kept.to(at::kFloat).neg().contiguous();
  auto order2 = at::empty({1, k}, d.options().dtype(at::kInt));
  auto sorted2 = at::empty({1, k}, d.options());
  auto ws3 = at::empty({(int64_t)dana_sort_desc_workspace_bytes(1, k) + 16}, d.options().dtype(at::kByte));
  check(dana_sort_desc(neg.data_ptr<float>(), 1, k, order2.data_ptr<int>(), sorted2.data_ptr<float>(), ws3.data_ptr(),
                       (size_t)ws3.numel(), stream_of(d)), "dana_sort_desc");
  return sorted2.view({k}).neg().to(at::kLong);
}

// shape / dtype / device contract shared by the RoI operators (ROIAlign.h:11-42, ROIPool.h:11-41: 4-D float maps,
// rois [R][5] float on the same device)
void check_map(const at::Tensor& t, const at::Tensor& rois, const char* who, const char* what) {
  TORCH_CHECK(t.dim() == 4 && t.scalar_type() == at::kFloat, who, ": ", what, " must be a 4-D float tensor");
  TORCH_CHECK(rois.dim() == 2 && rois.size(1) == 5 && rois.scalar_type() == at::kFloat, who, ": rois must be [R, 5] float");
  TORCH_CHECK(rois.device() == t.device(), who, ": rois and ", what, " must be on the same device");
}

at::Tensor roi_align_forward(const at::Tensor& input, const at::Tensor& rois, double spatial_scale, int64_t pooled_height,
                             int64_t pooled_width, int64_t sampling_ratio) {
  on_gpu(input, "input");
  on_gpu(rois, "rois");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
  const auto in = input.contiguous(), r = rois.contiguous();
  check_map(in, r, "roi_align_forward", "input");
  const int B = (int)in.size(0), C = (int)in.size(1), H = (int)in.size(2), W = (int)in.size(3), R = (int)r.size(0);
  auto out = at::empty({R, C, pooled_height, pooled_width}, in.options());
  if (R == 0) return out;
  check(dana_roi_align_forward(in.data_ptr<float>(), r.data_ptr<float>(), out.data_ptr<float>(), B, C, H, W, R,
                               (float)spatial_scale, (int)pooled_height, (int)pooled_width, (int)sampling_ratio,
                               DANA_LAYOUT_NCHW, 0, 0, nullptr, nullptr, 0, stream_of(in)), "dana_roi_align_forward");
  return out;
}

at::Tensor roi_align_backward(const at::Tensor& grad, const at::Tensor& rois, double spatial_scale, int64_t pooled_height,
                              int64_t pooled_width, int64_t batch_size, int64_t channels, int64_t height, int64_t width,
                              int64_t sampling_ratio) {
  on_gpu(grad, "grad");
  on_gpu(rois, "rois");
  check_map(grad, rois, "roi_align_backward", "grad");
  TORCH_CHECK(grad.size(0) == rois.size(0) && grad.size(1) == channels && grad.size(2) == pooled_height &&
              grad.size(3) == pooled_width, "roi_align_backward: grad must be [R, C, pooled_height, pooled_width]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(grad.device());
  const auto g = grad.contiguous(), r = rois.contiguous();
  auto gin = at::empty({batch_size, channels, height, width}, g.options());
  check(dana_roi_align_backward(g.data_ptr<float>(), r.data_ptr<float>(), gin.data_ptr<float>(), (int)batch_size,
                                (int)channels, (int)height, (int)width, (int)r.size(0), (float)spatial_scale,
                                (int)pooled_height, (int)pooled_width, (int)sampling_ratio, DANA_LAYOUT_NCHW,
                                stream_of(g)), "dana_roi_align_backward");
  return gin;
}

std::tuple<at::Tensor, at::Tensor> roi_pool_forward(const at::Tensor& input, const at::Tensor& rois, double spatial_scale,
                                                    int64_t pooled_height, int64_t pooled_width) {
  on_gpu(input, "input");
  on_gpu(rois, "rois");
  check_map(input, rois, "roi_pool_forward", "input");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(input.device());
  const auto in = input.contiguous(), r = rois.contiguous();
  const int B = (int)in.size(0), C = (int)in.size(1), H = (int)in.size(2), W = (int)in.size(3), R = (int)r.size(0);
  auto out = at::empty({R, C, pooled_height, pooled_width}, in.options());
  auto argmax = at::empty({R, C, pooled_height, pooled_width}, in.options().dtype(at::kInt));
  if (R == 0) return {out, argmax};
  check(dana_roi_pool_forward(in.data_ptr<float>(), r.data_ptr<float>(), out.data_ptr<float>(), argmax.data_ptr<int>(), B,
                              C, H, W, R, (float)spatial_scale, (int)pooled_height, (int)pooled_width, stream_of(in)),
        "dana_roi_pool_forward");
  return {out, argmax};
}

at::Tensor roi_pool_backward(const at::Tensor& grad, const at::Tensor& input, const at::Tensor& rois,
                             const at::Tensor& argmax, double spatial_scale, int64_t pooled_height, int64_t pooled_width,
                             int64_t batch_size, int64_t channels, int64_t height, int64_t width) {
  on_gpu(grad, "grad");
  on_gpu(rois, "rois");
  on_gpu(argmax, "argmax");
  check_map(grad, rois, "roi_pool_backward", "grad");
  TORCH_CHECK(argmax.scalar_type() == at::kInt && argmax.device() == grad.device() && argmax.sizes() == grad.sizes(),
              "roi_pool_backward: argmax must be the int32 tensor roi_pool_forward returned (same shape and device as grad)");
  TORCH_CHECK(grad.size(0) == rois.size(0) && grad.size(1) == channels, "roi_pool_backward: grad must be [R, C, ph, pw]");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(grad.device());
  const auto g = grad.contiguous(), r = rois.contiguous(), a = argmax.contiguous();
  auto gin = at::empty({batch_size, channels, height, width}, g.options());
  check(dana_roi_pool_backward(g.data_ptr<float>(), a.data_ptr<int>(), r.data_ptr<float>(), gin.data_ptr<float>(),
                               (int)batch_size, (int)channels, (int)height, (int)width, (int)r.size(0),
                               (int)pooled_height, (int)pooled_width, stream_of(g)), "dana_roi_pool_backward");
  (void)input;
  (void)spatial_scale;
  return gin;
}

// ---- differentiable forms (lib/model/roi_layers/roi_align.py:12-43, roi_pool.py:12-43) --------------------------------
struct RoIAlignFn : public torch::autograd::Function<RoIAlignFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& input, const at::Tensor& rois,
                            double spatial_scale, int64_t ph, int64_t pw, int64_t sampling_ratio) {
    ctx->save_for_backward({rois});
    ctx->saved_data["scale"] = spatial_scale;
    ctx->saved_data["ph"] = ph;
    ctx->saved_data["pw"] = pw;
    ctx->saved_data["sr"] = sampling_ratio;
    ctx->saved_data["shape"] = input.sizes().vec();
    at::AutoDispatchBelowADInplaceOrView guard;
    return roi_align_forward(input, rois, spatial_scale, ph, pw, sampling_ratio);
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    const auto rois = ctx->get_saved_variables()[0];
    const auto shape = ctx->saved_data["shape"].toIntVector();
    auto gin = roi_align_backward(grads[0], rois, ctx->saved_data["scale"].toDouble(), ctx->saved_data["ph"].toInt(),
                                  ctx->saved_data["pw"].toInt(), shape[0], shape[1], shape[2], shape[3],
                                  ctx->saved_data["sr"].toInt());
    return {gin, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

struct RoIPoolFn : public torch::autograd::Function<RoIPoolFn> {
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& input, const at::Tensor& rois,
                            double spatial_scale, int64_t ph, int64_t pw) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto res = roi_pool_forward(input, rois, spatial_scale, ph, pw);
    ctx->save_for_backward({input, rois, std::get<1>(res)});
    ctx->saved_data["scale"] = spatial_scale;
    ctx->saved_data["ph"] = ph;
    ctx->saved_data["pw"] = pw;
    return std::get<0>(res);
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const auto& input = saved[0];
    auto gin = roi_pool_backward(grads[0], input, saved[1], saved[2], ctx->saved_data["scale"].toDouble(),
                                 ctx->saved_data["ph"].toInt(), ctx->saved_data["pw"].toInt(), input.size(0),
                                 input.size(1), input.size(2), input.size(3));
    return {gin, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

at::Tensor roi_align(const at::Tensor& input, const at::Tensor& rois, double spatial_scale, int64_t ph, int64_t pw,
                     int64_t sampling_ratio) {
  return RoIAlignFn::apply(input, rois, spatial_scale, ph, pw, sampling_ratio);
}

at::Tensor roi_pool(const at::Tensor& input, const at::Tensor& rois, double spatial_scale, int64_t ph, int64_t pw) {
  return RoIPoolFn::apply(input, rois, spatial_scale, ph, pw);
}

}  // namespace

TORCH_LIBRARY(dana, m) {
  m.def("nms(Tensor dets, Tensor scores, float threshold) -> Tensor");
  m.def("roi_align_forward(Tensor input, Tensor rois, float spatial_scale, int pooled_height, int pooled_width, "
        "int sampling_ratio) -> Tensor");
  m.def("roi_align_backward(Tensor grad, Tensor rois, float spatial_scale, int pooled_height, int pooled_width, "
        "int batch_size, int channels, int height, int width, int sampling_ratio) -> Tensor");
  m.def("roi_pool_forward(Tensor input, Tensor rois, float spatial_scale, int pooled_height, int pooled_width) -> "
        "(Tensor, Tensor)");
  m.def("roi_pool_backward(Tensor grad, Tensor input, Tensor rois, Tensor argmax, float spatial_scale, "
        "int pooled_height, int pooled_width, int batch_size, int channels, int height, int width) -> Tensor");
  m.def("roi_align(Tensor input, Tensor rois, float spatial_scale, int pooled_height, int pooled_width, "
        "int sampling_ratio) -> Tensor");
  m.def("roi_pool(Tensor input, Tensor rois, float spatial_scale, int pooled_height, int pooled_width) -> Tensor");
}

// the raw five: plain kernels, no autograd formula of their own (CompositeExplicitAutograd: they run on whatever tensors
// they are given; the on_gpu() checks reject CPU tensors with the reference's wording)
TORCH_LIBRARY_IMPL(dana, CompositeExplicitAutograd, m) {
  m.impl("nms", &nms);
  m.impl("roi_align_forward", &roi_align_forward);
  m.impl("roi_align_backward", &roi_align_backward);
  m.impl("roi_pool_forward", &roi_pool_forward);
  m.impl("roi_pool_backward", &roi_pool_backward);
}

TORCH_LIBRARY_IMPL(dana, Autograd, m) {
  m.impl("roi_align", &roi_align);
  m.impl("roi_pool", &roi_pool);
}

// ... and below the Autograd key (torch.inference_mode(), or any context that excludes it): the plain forward
at::Tensor roi_align_plain(const at::Tensor& input, const at::Tensor& rois, double spatial_scale, int64_t ph, int64_t pw,
                           int64_t sampling_ratio) {
  return roi_align_forward(input, rois, spatial_scale, ph, pw, sampling_ratio);
}
at::Tensor roi_pool_plain(const at::Tensor& input, const at::Tensor& rois, double spatial_scale, int64_t ph, int64_t pw) {
  return std::get<0>(roi_pool_forward(input, rois, spatial_scale, ph, pw));
}
TORCH_LIBRARY_IMPL(dana, CompositeExplicitAutograd, m) {  // (every backend: CPU tensors get the reference's error text)
  m.impl("roi_align", &roi_align_plain);
  m.impl("roi_pool", &roi_pool_plain);
}

// torch_ext.cpp -- the PyTorch-ROCm operator library of the Pi-block hot path ("registered via TORCH_LIBRARY(percnn, m)
// as percnn::pi_step / pi_rollout (+ backward) + torch::autograd::Function").
//
// What the existing PeRCNN modules call in place of the per-step ATen sequence of RCNNCell.forward
// (DataDrivenModeling/2d_gs_rd/train_2drd.py:105-121) and of the time loop of RCNN.forward (train_2drd.py:162-190):
//
//   percnn::pi_step(Tensor h, Tensor params, str options="") -> Tensor                      (+ autograd)
//   percnn::pi_step_backward(Tensor h, Tensor params, Tensor g_out, str options="") -> (Tensor, Tensor)
//   percnn::pi_rollout(Tensor h0, Tensor params, int steps, str options="") -> Tensor        (+ autograd)
//   percnn::pi_rollout_backward(Tensor traj, Tensor params, Tensor g_traj, str options="") -> (Tensor, Tensor)
//
// dispatcher -> launcher: no Python / ctypes frame between `cell(h)` and the C-ABI of include/percnn_pi.h (libpercnn_pi.so,
// which this library links).  PyTorch supplies device memory, the current HIP stream and the autograd graph -- nothing else.
// FakeTensor implementations are registered from Python (percnn_amd/ops.py: torch.library.register_fake).
//
// The module part (PyInit_percnn_torch) is the EAGER fast path of a reference-style step loop (`for step in range(T): h, _ =
// cell(h)`, train_2drd.py:169-188), where the host side of a 100^2 step costs more than its 1.3 us kernel:
//   block_key(tensors, ...)  the cache key of RCNNCell.param_block (19 version counters + storage addresses) as one hash
//   step_nograd(h, P)        straight to the kernel, no dispatcher
//   cell_step(h, P, acc)     one autograd node per step whose backward ADDS its parameter-gradient sums to the shared
//                            double accumulator `acc` of the packed block (percnn_pi_step_bwd_* accumulates) and returns no
//                            gradient for P at all: the block's own node (functional.PackBlockFunction) delivers `acc` once
//                            per backward pass -- T-1 cast / add launches and T allocations less than one gradient per node.
#include <cstdlib>
#include <torch/extension.h>
#include <torch/library.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>
#include <torch/csrc/autograd/graph_task.h>

#include <mutex>
#include <string>
#include <tuple>

#include "percnn_pi.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

constexpr int64_t NPOLY = 36, NADV = 60;

[[noreturn]] void fail(int rc, const char* what)
{
    const char* msg;
    switch (rc) {
    case PERCNN_PI_EINVAL: msg = "invalid argument"; break;
    case PERCNN_PI_EWORKSPACE: msg = "workspace too small"; break;
    case PERCNN_PI_ETOOLARGE: msg = "grid too large for the step kernels' 32-bit offsets (a 2D field or a 3D plane of >= 4 GiB per species)"; break;
    case PERCNN_PI_EASYNC:
        msg = "an EARLIER call's persistent launch (resident forward or tile sweep) aborted on the device (its workgroups could not "
              "all be resident: another process / kernel holds CUs, or a CU mask is set) -- the results of that earlier "
              "persistent launch (forward or backward) are invalid; this call "
              "launched nothing (percnn_amd._lib.persist_status() tells where)";
        break;
    default: msg = nullptr;
    }
    if (msg) TORCH_CHECK(false, "percnn_amd: ", what, " failed: ", msg);
    TORCH_CHECK(false, "percnn_amd: ", what, " failed: hipError_t ", rc);
}
inline void check(int rc, const char* what) { if (rc != 0) fail(rc, what); }

inline void require(const Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), "percnn_amd: ", name, " must live on a HIP device (got ", t.device(), "); there is no CPU path");
    TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kDouble, "percnn_amd: ", name,
                " must be float32 or float64, got ", t.scalar_type());
}
inline void require_like(const Tensor& t, const Tensor& ref, const char* name)
{
    require(t, name);
    TORCH_CHECK(t.scalar_type() == ref.scalar_type(), "percnn_amd: ", name, " has dtype ", t.scalar_type(), ", expected ",
                ref.scalar_type());
    TORCH_CHECK(t.device() == ref.device(), "percnn_amd: ", name, " lives on ", t.device(), ", expected ", ref.device());
}
inline void check_state(const Tensor& h)
{
    TORCH_CHECK((h.dim() == 4 || h.dim() == 5) && h.size(0) == 1 && h.size(1) == 2,
                "percnn_amd: state must be [1,2,*S] (batch 1, two species), got ", h.sizes());
}

// hidden width encoded by the block length; 0 = pre-contracted polynomial block, -1 = advective block
inline int hc_of(const Tensor& P)
{
    const int64_t n = P.numel();
    if (P.dim() == 1 && n == NPOLY) return 0;
    if (P.dim() == 1 && n == NADV) return -1;
    const int64_t m = n - 16;
    TORCH_CHECK(P.dim() == 1 && m >= 22 && m % 2 == 0 && (m / 2 - 1) % 10 == 0, "percnn_amd: parameter block has ", n,
                " entries; expected 16 + 2*(10*hc+1)");
    return (int)((m / 2 - 1) / 10);
}

struct Shape {
    int64_t s[3];
    int ndim;
    explicit Shape(const Tensor& state, int first)     // spatial extents of `state`, starting at dimension `first`
    {
        ndim = (int)state.dim() - first;
        TORCH_CHECK(ndim == 2 || ndim == 3, "percnn_amd: 2 or 3 spatial dimensions");
        for (int i = 0; i < ndim; ++i) s[i] = state.size(first + i);
    }
};

inline void* stream_of(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
inline const char* opt_c(const std::string& o) { return o.empty() ? nullptr : o.c_str(); }

// ---- raw calls -----------------------------------------------------------------------------------------------------
Tensor step_fwd_raw(const Tensor& h, const Tensor& P, const std::string& options)
{
    const Shape sh(h, 2);
    Tensor out = at::empty_like(h);
    void* st = stream_of(h);
    const int hc = hc_of(P);
    int rc;
    if (h.scalar_type() == at::kFloat)
        rc = percnn_pi_step_fwd_opt_f32(h.const_data_ptr<float>(), out.mutable_data_ptr<float>(), P.const_data_ptr<float>(), hc,
                                        sh.ndim, sh.s, opt_c(options), st);
    else
        rc = percnn_pi_step_fwd_opt_f64(h.const_data_ptr<double>(), out.mutable_data_ptr<double>(), P.const_data_ptr<double>(), hc,
                                        sh.ndim, sh.s, opt_c(options), st);
    check(rc, "step_fwd");
    return out;
}

// workspace of one adjoint step (partial rows of the gradient sums): from the caching allocator, ~1 us
Tensor step_workspace(const Tensor& h, int hc, const Shape& sh)
{
    const size_t nbytes = percnn_pi_bwd_workspace_bytes(hc, sh.ndim, sh.s, (int)h.element_size());
    TORCH_CHECK(nbytes != 0, "percnn_amd: invalid problem shape");
    return at::empty({(int64_t)nbytes}, h.options().dtype(at::kByte));
}

// adjoint of one step; param_grad: double[np], ACCUMULATED
Tensor step_bwd_raw(const Tensor& h, const Tensor& g_out, const Tensor& P, Tensor& param_grad, const std::string& options)
{
    const Shape sh(h, 2);
    const int hc = hc_of(P);
    Tensor g_in = at::empty_like(h);
    Tensor ws = step_workspace(h, hc, sh);
    void* st = stream_of(h);
    int rc;
    if (h.scalar_type() == at::kFloat)
        rc = percnn_pi_step_bwd_opt_f32(h.const_data_ptr<float>(), g_out.const_data_ptr<float>(), nullptr,
                                        g_in.mutable_data_ptr<float>(), param_grad.mutable_data_ptr<double>(), ws.mutable_data_ptr(),
                                        (size_t)ws.numel(), P.const_data_ptr<float>(), hc, sh.ndim, sh.s, opt_c(options), st);
    else
        rc = percnn_pi_step_bwd_opt_f64(h.const_data_ptr<double>(), g_out.const_data_ptr<double>(), nullptr,
                                        g_in.mutable_data_ptr<double>(), param_grad.mutable_data_ptr<double>(), ws.mutable_data_ptr(),
                                        (size_t)ws.numel(), P.const_data_ptr<double>(), hc, sh.ndim, sh.s, opt_c(options), st);
    check(rc, "step_bwd");
    return g_in;
}

// ---- registered operators (HIP tensors carry the CUDA dispatch key on PyTorch-ROCm) -----------------------------------
Tensor pi_step_impl(const Tensor& h, const Tensor& params, std::string options)
{
    check_state(h);
    require(h, "h");
    require_like(params, h, "params");
    c10::hip::HIPGuard guard(h.device().index());
    return step_fwd_raw(h.contiguous(), params.contiguous(), options);
}

std::tuple<Tensor, Tensor> pi_step_backward_impl(const Tensor& h, const Tensor& params, const Tensor& g_out, std::string options)
{
    check_state(h);
    require(h, "h");
    require_like(params, h, "params");
    require_like(g_out, h, "g_out");
    c10::hip::HIPGuard guard(h.device().index());
    const Tensor P = params.contiguous();
    Tensor pg = at::zeros({P.numel()}, h.options().dtype(at::kDouble));
    Tensor g_in = step_bwd_raw(h.contiguous(), g_out.contiguous(), P, pg, options);
    return {g_in, pg.to(P.scalar_type())};
}

Tensor pi_rollout_impl(const Tensor& h0, const Tensor& params, int64_t steps, std::string options)
{
    check_state(h0);
    require(h0, "h0");
    require_like(params, h0, "params");
    TORCH_CHECK(steps >= 0, "percnn_amd: steps must be >= 0");
    c10::hip::HIPGuard guard(h0.device().index());
    const Tensor P = params.contiguous();
    auto sizes = h0.sizes().vec();
    sizes[0] = steps + 1;
    Tensor traj = at::empty(sizes, h0.options());
    traj.select(0, 0).copy_(h0.select(0, 0));
    const Shape sh(traj, 2);
    void* st = stream_of(traj);
    const int hc = hc_of(P);
    int rc;
    if (traj.scalar_type() == at::kFloat)
        rc = percnn_pi_rollout_fwd_opt_f32(traj.mutable_data_ptr<float>(), P.const_data_ptr<float>(), hc, sh.ndim, sh.s, (int)steps,
                                           opt_c(options), st);
    else
        rc = percnn_pi_rollout_fwd_opt_f64(traj.mutable_data_ptr<double>(), P.const_data_ptr<double>(), hc, sh.ndim, sh.s, (int)steps,
                                           opt_c(options), st);
    check(rc, "rollout_fwd");
    return traj;
}

std::tuple<Tensor, Tensor> pi_rollout_backward_impl(const Tensor& traj, const Tensor& params, const Tensor& g_traj, std::string options)
{
    require(traj, "traj");
    require_like(params, traj, "params");
    require_like(g_traj, traj, "g_traj");
    TORCH_CHECK(traj.is_contiguous(), "percnn_amd: traj must be contiguous");
    TORCH_CHECK(traj.dim() >= 4 && traj.size(1) == 2 && traj.size(0) >= 1, "percnn_amd: traj must be [T+1,2,*S]");
    c10::hip::HIPGuard guard(traj.device().index());
    const Tensor P = params.contiguous(), g = g_traj.contiguous();
    const Shape sh(traj, 2);
    const int hc = hc_of(P), T = (int)traj.size(0) - 1;
    auto s1 = traj.sizes().vec();
    s1[0] = 1;
    Tensor g_h0 = at::empty(s1, traj.options());
    Tensor pg = at::zeros({P.numel()}, traj.options().dtype(at::kDouble));
    const size_t nbytes = percnn_pi_rollout_bwd_workspace_bytes(hc, sh.ndim, sh.s, T, (int)traj.element_size());
    TORCH_CHECK(nbytes != 0, "percnn_amd: invalid problem shape");
    Tensor ws = at::empty({(int64_t)nbytes}, traj.options().dtype(at::kByte));
    void* st = stream_of(traj);
    int rc;
    if (traj.scalar_type() == at::kFloat)
        rc = percnn_pi_rollout_bwd_opt_f32(traj.const_data_ptr<float>(), g.const_data_ptr<float>(), nullptr, g_h0.mutable_data_ptr<float>(),
                                           pg.mutable_data_ptr<double>(), ws.mutable_data_ptr(), (size_t)ws.numel(),
                                           P.const_data_ptr<float>(), hc, sh.ndim, sh.s, T, opt_c(options), st);
    else
        rc = percnn_pi_rollout_bwd_opt_f64(traj.const_data_ptr<double>(), g.const_data_ptr<double>(), nullptr,
                                           g_h0.mutable_data_ptr<double>(), pg.mutable_data_ptr<double>(), ws.mutable_data_ptr(),
                                           (size_t)ws.numel(), P.const_data_ptr<double>(), hc, sh.ndim, sh.s, T, opt_c(options), st);
    check(rc, "rollout_bwd");
    return {g_h0, pg.to(P.scalar_type())};
}

// ---- autograd formulas of the registered operators ----------------------------------------------------------------------
struct PiStepFn : public torch::autograd::Function<PiStepFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& h, const Tensor& params, std::string options)
    {
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("percnn::pi_step", "")
                             .typed<Tensor(const Tensor&, const Tensor&, std::string)>();
        Tensor out = op.call(h, params, options);
        ctx->save_for_backward({h, params});
        ctx->saved_data["options"] = options;
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("percnn::pi_step_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, std::string)>();
        auto [g_in, g_p] = op.call(saved[0], saved[1], grads[0], ctx->saved_data["options"].toStringRef());
        return {g_in, g_p, Tensor()};
    }
};
Tensor pi_step_autograd(const Tensor& h, const Tensor& params, std::string options) { return PiStepFn::apply(h, params, options); }

struct PiRolloutFn : public torch::autograd::Function<PiRolloutFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& h0, const Tensor& params, int64_t steps, std::string options)
    {
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("percnn::pi_rollout", "")
                             .typed<Tensor(const Tensor&, const Tensor&, int64_t, std::string)>();
        Tensor traj = op.call(h0, params, steps, options);
        ctx->save_for_backward({traj, params});
        ctx->saved_data["options"] = options;
        return traj;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("percnn::pi_rollout_backward", "")
                             .typed<std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const Tensor&, std::string)>();
        auto [g_h0, g_p] = op.call(saved[0], saved[1], grads[0], ctx->saved_data["options"].toStringRef());
        return {g_h0, g_p, Tensor(), Tensor()};
    }
};
Tensor pi_rollout_autograd(const Tensor& h0, const Tensor& params, int64_t steps, std::string options)
{
    return PiRolloutFn::apply(h0, params, steps, options);
}

// ---- eager fast path of a reference-style step loop -----------------------------------------------------------------------
// Native state of ONE packed parameter block (RCNNCell.param_block caches the block together with these objects).
//
// GradSink -- shared parameter-gradient accumulator.  The autograd nodes a step loop creates do NOT return a gradient for the
//   block: single-step nodes leave their sums in the partial rows of ONE workspace (percnn_pi_step_bwd_rows_*: no reset, no
//   reduction launch per step), group nodes add theirs to `acc` directly (percnn_pi_rollout_bwd_* accumulates); the block's own
//   node (functional.PackBlockFunction) calls take(): one reduction launch, one cast.  A backward pass is identified by
//   autograd's graph-task id: sums left behind by a pass that never reached the block's node (torch.autograd.grad(loss, [h0])
//   prunes it) are dropped by the next pass, never delivered to it.
struct GradSink : torch::CustomClassHolder {
    std::mutex mu;
    Tensor acc;                 // double[np_max]
    Tensor ws;                  // workspace whose partial rows hold sums of the current pass
    int64_t task = -1;          // graph task the sums belong to (-1: none; acc is zero, rows are clean)
    bool rows_dirty = false, acc_dirty = false;
    int ws_hc = 0, ws_ndim = 0;
    int64_t ws_shape[3] = {0, 0, 0};
    at::ScalarType ws_dtype = at::kFloat;

    explicit GradSink(Tensor a) : acc(std::move(a)) {}

    bool same_problem(const Tensor& h, int hc, const Shape& sh) const
    {
        if (!ws.defined() || ws_hc != hc || ws_ndim != sh.ndim || ws_dtype != h.scalar_type() || ws.device() != h.device()) return false;
        for (int i = 0; i < sh.ndim; ++i) if (ws_shape[i] != sh.s[i]) return false;
        return true;
    }
    void flush_rows(void* st)                                // rows -> acc (one launch); the next step launch resets the rows
    {
        if (!rows_dirty) return;
        int rc;
        if (ws_dtype == at::kFloat)
            rc = percnn_pi_bwd_rows_finish_f32(ws.mutable_data_ptr(), (size_t)ws.numel(), ws_hc, ws_ndim, ws_shape, acc.mutable_data_ptr<double>(), st);
        else
            rc = percnn_pi_bwd_rows_finish_f64(ws.mutable_data_ptr(), (size_t)ws.numel(), ws_hc, ws_ndim, ws_shape, acc.mutable_data_ptr<double>(), st);
        check(rc, "bwd_rows_finish");
        rows_dirty = false;
        acc_dirty = true;
    }
    void drop()                                             // forget the sums of a pass nobody collected
    {
        if (acc_dirty) acc.zero_();
        acc_dirty = rows_dirty = false;
        task = -1;
    }
    void enter(int64_t current)
    {
        if (task != current) { if (task != -1) drop(); task = current; }
    }
    // adjoint of one step of the pass `current`, sums into the rows
    Tensor step_bwd(const Tensor& h, const Tensor& g, const Tensor& P, int64_t current)
    {
        std::lock_guard<std::mutex> lk(mu);
        const Shape sh(h, 2);
        const int hc = hc_of(P);
        void* st = stream_of(h);
        enter(current);
        if (!same_problem(h, hc, sh)) {
            flush_rows(st);
            ws = step_workspace(h, hc, sh);
            ws_hc = hc; ws_ndim = sh.ndim; ws_dtype = h.scalar_type();
            for (int i = 0; i < 3; ++i) ws_shape[i] = i < sh.ndim ? sh.s[i] : 0;
        }
        Tensor g_in = at::empty_like(h);
        const int flags = rows_dirty ? PERCNN_PI_NO_RESET : 0;
        int rc;
        if (h.scalar_type() == at::kFloat)
            rc = percnn_pi_step_bwd_rows_f32(h.const_data_ptr<float>(), g.const_data_ptr<float>(), nullptr, g_in.mutable_data_ptr<float>(),
                                             ws.mutable_data_ptr(), (size_t)ws.numel(), P.const_data_ptr<float>(), hc, sh.ndim, sh.s, flags, st);
        else
            rc = percnn_pi_step_bwd_rows_f64(h.const_data_ptr<double>(), g.const_data_ptr<double>(), nullptr, g_in.mutable_data_ptr<double>(),
                                             ws.mutable_data_ptr(), (size_t)ws.numel(), P.const_data_ptr<double>(), hc, sh.ndim, sh.s, flags, st);
        check(rc, "step_bwd");
        rows_dirty = true;
        return g_in;
    }
    // called by the block's node: the sums of THIS pass as a tensor of `like`'s dtype (or undefined), everything back to zero
    Tensor take(int64_t current, const Tensor& like)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (task == -1) return Tensor();
        if (task != current) { drop(); return Tensor(); }
        c10::hip::HIPGuard guard(acc.device().index());
        flush_rows(c10::hip::getCurrentHIPStream(acc.device().index()).stream());
        Tensor out = acc.to(like.scalar_type(), /*non_blocking=*/false, /*copy=*/true);   // acc is zeroed next
        drop();
        return out;
    }
};

int64_t current_task() { return (int64_t)torch::autograd::get_current_graph_task_id(); }

struct BlockState;
Tensor single_step_recorded(const Tensor& h, const Tensor& P, const c10::intrusive_ptr<GradSink>& sink, BlockState* bs);
variable_list group_recorded(const Tensor& h, const Tensor& P, BlockState* bs, int64_t want);

// BlockState -- speculative steps.  `for step in range(T): h, _ = cell(h)` hands each output back as the next input.  When the
//   input IS the tensor the previous call returned (same TensorImpl, unmodified, same block, same stream, same grad mode), the
//   following states are already determined: the call that notices computes the next 4 / 8 / 16 steps with ONE call of the fused
//   rollout (percnn_pi_rollout_fwd_*: the temporally blocked kernels) into a chunk of frames, and the next calls return those
//   frames without launching anything.  Bit-identical to step-by-step (the fused kernels are bit-identical to the direct ones,
//   asserted by the parity tests); anything else -- another input, an input modified in place, another stream, a new block --
//   is a plain single step.  Frames are separate tensors that share the chunk's storage (like the frames RCNN.forward
//   returns); grids above 2^21 scalars per state are never speculated on.
//   While autograd records, such a group of steps is ONE node with one output per frame (GroupStepFn): the chain between its
//   frames is internal to the node, so neither T nodes nor T gradient additions by the engine, and its backward is ONE call of
//   the fused sweep over the group (percnn_pi_rollout_bwd_*, frames without a gradient masked).
struct BlockState : torch::CustomClassHolder {
    std::mutex mu;
    c10::intrusive_ptr<GradSink> sink;
    Tensor chunk;               // [L, 2, *S] frames
    std::vector<Tensor> frames; // frames[i]: [1, 2, *S] tensor on chunk's storage (created when frame i is computed)
    int64_t valid = -1;         // frames[0 .. valid] hold enqueued results
    int64_t next = -1;          // frame a hit returns
    int depth = 0;              // steps of the last speculative launch (0: none yet)
    int64_t chain_len = 0;      // frames handed out since the current chain of steps began (a call that matched nothing begins one)
    int64_t last_chain_len = 0; // ... of the chain before it: a training loop runs the same number of steps every iteration, so the
                                // groups of this chain are cut to end where the last one ended (round 6)
    bool recorded = false;      // the frames beyond `next` belong to an autograd node
    c10::TensorImpl* last_out = nullptr;
    uint32_t last_version = 0;
    c10::TensorImpl* p_impl = nullptr;
    uint32_t p_version = 0;
    void* stream = nullptr;
    bool speculate = true;
    int64_t spec_launches = 0, spec_hits = 0;
    // the cached block this state belongs to and the key it was packed under (RCNNCell.param_block; fast_forward below)
    Tensor block;
    int64_t key = -1;

    explicit BlockState(Tensor a) : sink(c10::make_intrusive<GradSink>(std::move(a))) {}

    void forget()
    {
        frames.clear(); chunk = Tensor();
        valid = next = -1; depth = 0; last_out = nullptr; p_impl = nullptr; chain_len = last_chain_len = 0;
    }
    Tensor frame_tensor(int64_t i) const                   // a NON-view tensor on the chunk's storage: own version counter
    {
        auto sizes = chunk.sizes().vec();
        sizes[0] = 1;
        Tensor t = at::empty({0}, chunk.options());
        t.set_(chunk.storage(), chunk.storage_offset() + i * chunk.stride(0), sizes, chunk.strides());
        return t;
    }
    Tensor returned(int64_t i, void* st, const Tensor& P)
    {
        const Tensor& out = frames[(size_t)i];
        last_out = out.unsafeGetTensorImpl();
        last_version = out._version();
        p_impl = P.unsafeGetTensorImpl();
        p_version = P._version();
        stream = st;
        next = i + 1;
        ++chain_len;
        return out;
    }
    // frames of one allocation.  Every frame a caller keeps pins its whole chunk, so chunks stay small: 32 MiB at most, and TWO
    // frames for the single step of a call that matched nothing (a loop that never hands an output back -- `h = f(cell(h)[0])` --
    // would otherwise pin a full chunk per step); the first call that does match moves on to a full chunk (one frame copy).
    void new_chunk(const Tensor& h, bool single = false)
    {
        const int64_t frame_bytes = h.numel() * (int64_t)h.element_size();
        int64_t L = (int64_t(32) << 20) / (frame_bytes > 0 ? frame_bytes : 1);
        L = single ? 2 : (L < 6 ? 6 : (L > 64 ? 64 : L));
        auto sizes = h.sizes().vec();
        sizes[0] = L;
        chunk = at::empty(sizes, h.options());
        frames.assign((size_t)L, Tensor());
        valid = -1;
    }
    // the launch of a group: frames from+1 .. from+want of the chunk (from = the frame that holds h's values); no autograd here
    variable_list launch_group(const Tensor& h, const Tensor& P, int64_t from, int64_t want)
    {
        const Shape sh(h, 2);
        const int hc = hc_of(P);
        void* st = stream_of(h);
        char* base = static_cast<char*>(chunk.mutable_data_ptr()) + from * chunk.stride(0) * (int64_t)chunk.element_size();
        int rc;
        if (h.scalar_type() == at::kFloat)
            rc = percnn_pi_rollout_fwd_opt_f32(reinterpret_cast<float*>(base), P.const_data_ptr<float>(), hc, sh.ndim, sh.s, (int)want, nullptr, st);
        else
            rc = percnn_pi_rollout_fwd_opt_f64(reinterpret_cast<double*>(base), P.const_data_ptr<double>(), hc, sh.ndim, sh.s, (int)want, nullptr, st);
        check(rc, "rollout_fwd");
        variable_list outs;
        outs.reserve((size_t)want);
        for (int64_t i = from + 1; i <= from + want; ++i) outs.push_back(frame_tensor(i));
        return outs;
    }
    Tensor single_step(const Tensor& h, const Tensor& P)   // h -> frame 1 of a fresh chunk; no autograd here
    {
        const Shape sh(h, 2);
        const int hc = hc_of(P);
        void* st = stream_of(h);
        new_chunk(h, true);
        char* o = static_cast<char*>(chunk.mutable_data_ptr()) + chunk.stride(0) * (int64_t)chunk.element_size();
        int rc;
        if (h.scalar_type() == at::kFloat)
            rc = percnn_pi_step_fwd_opt_f32(h.const_data_ptr<float>(), reinterpret_cast<float*>(o), P.const_data_ptr<float>(), hc, sh.ndim,
                                            sh.s, nullptr, st);
        else
            rc = percnn_pi_step_fwd_opt_f64(h.const_data_ptr<double>(), reinterpret_cast<double*>(o), P.const_data_ptr<double>(), hc, sh.ndim,
                                            sh.s, nullptr, st);
        check(rc, "step_fwd");
        return frame_tensor(1);
    }
    // one forward step of a loop; h, P contiguous; record: autograd is recording for this call
    Tensor step(const Tensor& h, const Tensor& P, bool record)
    {
        std::lock_guard<std::mutex> lk(mu);
        void* st = stream_of(h);
        const bool small = h.numel() <= (int64_t(2) << 20);
        if (!speculate || !small) {
            if (record) return single_step_recorded(h, P, sink, nullptr);
            return step_fwd_raw(h, P, std::string());
        }
        const bool hit = last_out != nullptr && h.unsafeGetTensorImpl() == last_out && h._version() == last_version &&
                         P.unsafeGetTensorImpl() == p_impl && P._version() == p_version && st == stream && chunk.defined();
        if (hit && next <= valid && record == recorded) { ++spec_hits; return returned(next, st, P); }
        if (hit && next > valid) {
            // the input is the newest frame we hold: the loop pattern.  Compute the next `want` steps in one fused call.
            const int64_t L = chunk.size(0);
            // 4, 8, 16, 16, ... steps per launch.  Deeper groups (32 / 64 steps, with or without the resident kernels behind them)
            // were measured in round 6 and LOSE: the loop is bound by the 200 Python -> dispatcher calls, not by the 13 group
            // launches, and longer sweeps made the backward slower (profiles/r06_step_loop_speculation_depth.txt).
            // A chain as long as the previous one is not speculated past its end.
            int64_t want = depth == 0 ? 4 : (depth < 16 ? 2 * depth : 16);
            if (last_chain_len > chain_len && want > last_chain_len - chain_len) want = last_chain_len - chain_len;
            int64_t from = next - 1;                         // frame index of h
            if (from + 1 >= L) {                             // chunk exhausted: its last frame becomes frame 0 of a new one
                const Tensor keep = h;
                new_chunk(h);
                chunk.select(0, 0).copy_(keep.select(0, 0));
                frames[0] = keep;                            // (never returned again; keeps the indexing simple)
                from = 0;
            }
            if (from + want >= L) want = L - 1 - from;
            group_from = from;
            variable_list outs = record ? group_recorded(h, P, this, want) : launch_group(h, P, from, want);
            for (int64_t i = 0; i < want; ++i) frames[(size_t)(from + 1 + i)] = outs[(size_t)i];
            valid = from + want;
            depth = (int)want;
            recorded = record;
            ++spec_launches;
            return returned(from + 1, st, P);
        }
        // anything else: a plain single step, written into frame 1 of a fresh chunk so that the next call can recognise its output
        depth = 0;
        if (chain_len > 1) last_chain_len = chain_len;      // (single calls that never chain leave the estimate alone)
        chain_len = 0;
        Tensor out = record ? single_step_recorded(h, P, sink, this) : single_step(h, P);
        frames[1] = out;
        valid = 1;
        recorded = record;
        return returned(1, st, P);
    }
    int64_t group_from = 0;     // (argument of the group launch in flight: see GroupStepFn::forward)
};

// one step = one node; its backward leaves the parameter-gradient sums in the sink's rows
struct CellStepFn : public torch::autograd::Function<CellStepFn> {
    static Tensor forward(AutogradContext* ctx, const Tensor& h, const Tensor& params, const c10::intrusive_ptr<GradSink>& sink,
                          int64_t bs_ptr)
    {
        BlockState* bs = reinterpret_cast<BlockState*>(bs_ptr);
        Tensor out = bs ? bs->single_step(h, params) : step_fwd_raw(h, params, std::string());
        ctx->save_for_backward({h, params});
        ctx->saved_data["sink"] = c10::IValue(sink);
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const Tensor& h = saved[0];
        const Tensor& P = saved[1];
        if (!grads[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
        c10::hip::HIPGuard guard(h.device().index());
        const Tensor g = grads[0].contiguous();
        if (ctx->needs_input_grad(1)) {
            auto sink = ctx->saved_data["sink"].toCustomClass<GradSink>();
            return {sink->step_bwd(h, g, P, current_task()), Tensor(), Tensor(), Tensor()};
        }
        // nobody asked for dL/dparams in this pass: the sums go to a scratch block
        Tensor scratch = at::zeros({P.numel()}, h.options().dtype(at::kDouble));
        return {step_bwd_raw(h, g, P, scratch, std::string()), Tensor(), Tensor(), Tensor()};
    }
};

// a group of speculated steps = one node with one output per frame
struct GroupStepFn : public torch::autograd::Function<GroupStepFn> {
    static variable_list forward(AutogradContext* ctx, const Tensor& h, const Tensor& params, const c10::intrusive_ptr<GradSink>& sink,
                                 int64_t bs_ptr, int64_t want)
    {
        BlockState* bs = reinterpret_cast<BlockState*>(bs_ptr);
        const int64_t from = bs->group_from;
        variable_list outs = bs->launch_group(h, params, from, want);
        variable_list keep = {h, params};
        keep.insert(keep.end(), outs.begin(), outs.end());
        ctx->save_for_backward(keep);                       // (outputs included: an in-place edit of a frame is caught at unpack)
        ctx->saved_data["sink"] = c10::IValue(sink);
        ctx->saved_data["base"] = bs->frame_tensor(from);   // frame `from` of the chunk: h's values, contiguous with the outputs
        return outs;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const Tensor& h = saved[0];
        const Tensor& P = saved[1];
        const Tensor base = ctx->saved_data["base"].toTensor();
        int64_t T = 0;
        for (int64_t k = (int64_t)grads.size(); k >= 1; --k)
            if (grads[(size_t)k - 1].defined()) { T = k; break; }
        if (T == 0) return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
        c10::hip::HIPGuard guard(h.device().index());
        const Shape sh(h, 2);
        const int hc = hc_of(P);
        void* st = stream_of(h);
        const int64_t frame_elems = h.numel(), esz = (int64_t)h.element_size(), frame_bytes = frame_elems * esz;
        // dL/dtraj of the group: the gradients of frames 1..T.  After torch.cat, frames 1..T-1 are consecutive slices of ONE
        // buffer (CatBackward narrows its incoming gradient) and the sweep reads them in place; frame T -- which also fed the next
        // node, so the engine summed two contributions into a tensor of its own -- is passed separately (percnn_pi_rollout_bwd_top_*);
        // frames without a gradient are masked.
        std::vector<unsigned char> mask((size_t)T + 1, 0);
        mask[(size_t)T] = 1;
        const Tensor g_top = grads[(size_t)T - 1].contiguous();
        const char* first = nullptr;
        int64_t kf = 0;
        bool inplace = true;
        for (int64_t k = 1; k < T; ++k) {
            const Tensor& g = grads[(size_t)k - 1];
            if (!g.defined()) continue;
            mask[(size_t)k] = 1;
            if (!g.is_contiguous() || g.scalar_type() != h.scalar_type() || g.device() != h.device()) { inplace = false; continue; }
            const char* p = static_cast<const char*>(g.const_data_ptr());
            if (!first) { first = p; kf = k; }
            else if (p != first + (k - kf) * frame_bytes) inplace = false;
        }
        Tensor gbuf;
        const char* gptr;
        if (!first) {
            gptr = static_cast<const char*>(g_top.const_data_ptr());    // nothing below the top frame is read
        } else if (inplace) {
            gptr = first - kf * frame_bytes;                // (frames below kf are masked: never dereferenced there)
        } else {
            auto sizes = h.sizes().vec();
            sizes[0] = T;
            gbuf = at::empty(sizes, h.options());
            for (int64_t k = 1; k < T; ++k)
                if (mask[(size_t)k]) gbuf.select(0, k).copy_(grads[(size_t)k - 1].select(0, 0));
            gptr = static_cast<const char*>(gbuf.const_data_ptr());
        }
        Tensor g_h = at::empty_like(h);
        const size_t nbytes = percnn_pi_rollout_bwd_workspace_bytes(hc, sh.ndim, sh.s, (int)T, (int)esz);
        TORCH_CHECK(nbytes != 0, "percnn_amd: invalid problem shape");
        Tensor ws = at::empty({(int64_t)nbytes}, h.options().dtype(at::kByte));
        auto sink = ctx->saved_data["sink"].toCustomClass<GradSink>();
        Tensor pg;
        std::unique_lock<std::mutex> lk(sink->mu, std::defer_lock);
        if (ctx->needs_input_grad(1)) {
            lk.lock();
            sink->enter(current_task());
            pg = sink->acc;
            sink->acc_dirty = true;
        } else {
            pg = at::zeros({P.numel()}, h.options().dtype(at::kDouble));
        }
        // (short sweeps: the launch-per-group tile sweep; the persistent flavour pays a host handshake per call)
        const char* sweep_opts = "tile_persist=0";
        int rc;
        if (h.scalar_type() == at::kFloat)
            rc = percnn_pi_rollout_bwd_top_f32(base.const_data_ptr<float>(), reinterpret_cast<const float*>(gptr),
                                               g_top.const_data_ptr<float>(), mask.data(), g_h.mutable_data_ptr<float>(),
                                               pg.mutable_data_ptr<double>(), ws.mutable_data_ptr(), nbytes, P.const_data_ptr<float>(), hc,
                                               sh.ndim, sh.s, (int)T, sweep_opts, st);
        else
            rc = percnn_pi_rollout_bwd_top_f64(base.const_data_ptr<double>(), reinterpret_cast<const double*>(gptr),
                                               g_top.const_data_ptr<double>(), mask.data(), g_h.mutable_data_ptr<double>(),
                                               pg.mutable_data_ptr<double>(), ws.mutable_data_ptr(), nbytes, P.const_data_ptr<double>(), hc,
                                               sh.ndim, sh.s, (int)T, sweep_opts, st);
        check(rc, "rollout_bwd");
        return {g_h, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

Tensor single_step_recorded(const Tensor& h, const Tensor& P, const c10::intrusive_ptr<GradSink>& sink, BlockState* bs)
{
    return CellStepFn::apply(h, P, sink, (int64_t)reinterpret_cast<intptr_t>(bs));
}
variable_list group_recorded(const Tensor& h, const Tensor& P, BlockState* bs, int64_t want)
{
    return GroupStepFn::apply(h, P, bs->sink, (int64_t)reinterpret_cast<intptr_t>(bs), want);
}

inline void check_cell_args(const Tensor& h, const Tensor& params)
{
    check_state(h);
    require(h, "h");
    require_like(params, h, "params");
}

Tensor cell_step(const Tensor& h, const Tensor& params, const c10::intrusive_ptr<BlockState>& bs)
{
    check_cell_args(h, params);
    c10::hip::OptionalHIPGuard guard;
    if (c10::hip::current_device() != h.device().index()) guard.set_index(h.device().index());
    return bs->step(h.contiguous(), params.contiguous(), true);
}

Tensor step_nograd(const Tensor& h, const Tensor& params, const c10::optional<c10::intrusive_ptr<BlockState>>& bs)
{
    check_cell_args(h, params);
    at::NoGradGuard ng;
    c10::hip::OptionalHIPGuard guard;
    if (c10::hip::current_device() != h.device().index()) guard.set_index(h.device().index());
    if (bs.has_value() && h.is_contiguous() && params.is_contiguous()) return (*bs)->step(h, params, false);
    return step_fwd_raw(h.contiguous(), params.contiguous(), std::string());
}

// FNV-1a over (version counter, storage address) of every tensor of a Python list: RCNNCell.param_block's cache key without 38
// Python-level calls (the list is read in place: no vector of tensors is built)
int64_t block_key(const py::list& tensors)
{
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        for (int i = 0; i < 8; ++i) { hsh ^= (v >> (8 * i)) & 0xFFu; hsh *= 1099511628211ull; }
    };
    const Py_ssize_t n = PyList_GET_SIZE(tensors.ptr());
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* o = PyList_GET_ITEM(tensors.ptr(), i);
        TORCH_CHECK(THPVariable_Check(o), "block_key: a list of tensors");
        const Tensor& t = THPVariable_Unpack(o);
        mix((uint64_t)t._version());
        mix((uint64_t)reinterpret_cast<uintptr_t>(t.unsafeGetTensorImpl()->unsafe_storage().unsafeGetStorageImpl()->data()) +
            (uint64_t)t.storage_offset());
    }
    return (int64_t)(hsh & 0x7FFFFFFFFFFFFFFFull);
}

// The same key straight from the module tree: `src` lists, per tensor, (dict, module name or None, parameter name) -- the
// cell's own `_parameters` (name None) or its `_modules` + the name of the sub-module whose `_parameters` holds the tensor.  Hashes
// the identity of every sub-module and tensor OBJECT as well, so parameter surgery and replaced sub-modules change the key
// (one call per time step of a reference-style loop instead of ~60 Python-level lookups).
int64_t block_key_src(const py::list& src)
{
    static PyObject* params_attr = PyUnicode_InternFromString("_parameters");
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        for (int i = 0; i < 8; ++i) { hsh ^= (v >> (8 * i)) & 0xFFu; hsh *= 1099511628211ull; }
    };
    const Py_ssize_t n = PyList_GET_SIZE(src.ptr());
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* e = PyList_GET_ITEM(src.ptr(), i);
        TORCH_CHECK(PyTuple_Check(e) && PyTuple_GET_SIZE(e) == 3, "block_key_src: (dict, module name | None, parameter name) tuples");
        PyObject* d = PyTuple_GET_ITEM(e, 0);
        PyObject* mname = PyTuple_GET_ITEM(e, 1);
        PyObject* pname = PyTuple_GET_ITEM(e, 2);
        py::object holder;                                  // keeps the sub-module's _parameters alive while we look
        if (mname != Py_None) {
            PyObject* mod = PyDict_GetItemWithError(d, mname);
            if (!mod) return -1;
            mix((uint64_t)reinterpret_cast<uintptr_t>(mod));
            holder = py::reinterpret_steal<py::object>(PyObject_GetAttr(mod, params_attr));
            if (!holder) { PyErr_Clear(); return -1; }
            d = holder.ptr();
        }
        PyObject* o = PyDict_Check(d) ? PyDict_GetItemWithError(d, pname) : nullptr;
        if (!o || !THPVariable_Check(o)) return -1;
        const Tensor& t = THPVariable_Unpack(o);
        mix((uint64_t)reinterpret_cast<uintptr_t>(o));
        mix((uint64_t)t._version());
        mix((uint64_t)reinterpret_cast<uintptr_t>(t.unsafeGetTensorImpl()->unsafe_storage().unsafeGetStorageImpl()->data()) +
            (uint64_t)t.storage_offset());
    }
    return (int64_t)(hsh & 0x7FFFFFFFFFFFFFFFull);
}

// RCNNCell._block_key: the tensors (block_key_src over cell._pack_src_list) + everything else a packed block depends on --
// dt, reaction mode, diffusion kind, guard settings, speculation switch (hash of the attribute VALUES) and whether autograd
// records.  -1: something is missing or unhashable (the Python side then takes its slow path).
int64_t cell_key(PyObject* cell_dict)
{
    static PyObject* names[8] = {nullptr};
    if (!names[0]) {
        const char* n[8] = {"_pack_src_list", "dt", "reaction", "diffusion", "poly_guard", "state_bound", "poly_guard_max", "speculate"};
        for (int i = 0; i < 8; ++i) names[i] = PyUnicode_InternFromString(n[i]);
    }
    PyObject* src = PyDict_GetItemWithError(cell_dict, names[0]);
    if (!src || !PyList_Check(src)) return -1;
    const int64_t tk = block_key_src(py::reinterpret_borrow<py::list>(src));
    if (tk < 0) return -1;
    uint64_t hsh = (uint64_t)tk ^ (at::GradMode::is_enabled() ? 0x9E3779B97F4A7C15ull : 0ull);
    for (int i = 1; i < 8; ++i) {
        PyObject* v = PyDict_GetItemWithError(cell_dict, names[i]);
        if (!v) return -1;
        const Py_hash_t hv = PyObject_Hash(v);
        if (hv == -1 && PyErr_Occurred()) { PyErr_Clear(); return -1; }
        hsh = (hsh ^ (uint64_t)hv) * 1099511628211ull + (uint64_t)i;
    }
    return (int64_t)(hsh & 0x7FFFFFFFFFFFFFFFull);
}

// RCNNCell.forward's hit path in one call: validate the cached block against the module tree, then the step (speculated frame,
// single launch, or one autograd node).  Returns None when the cache does not apply -- the Python side repacks and comes back.
py::object fast_forward(const py::object& cell, const Tensor& h)
{
    PyObject** dp = _PyObject_GetDictPtr(cell.ptr());
    if (!dp || !*dp) return py::none();
    static PyObject* st_name = PyUnicode_InternFromString("_block_acc");
    PyObject* so = PyDict_GetItemWithError(*dp, st_name);
    if (!so || so == Py_None) return py::none();
    c10::intrusive_ptr<BlockState> bs;
    try { bs = py::cast<c10::intrusive_ptr<BlockState>>(py::handle(so)); } catch (const py::cast_error&) { return py::none(); }
    if (bs->key < 0 || !bs->block.defined() || cell_key(*dp) != bs->key) return py::none();
    const Tensor& P = bs->block;
    if (at::GradMode::is_enabled() && (h.requires_grad() || P.requires_grad())) return py::cast(cell_step(h, P, bs));
    return py::cast(step_nograd(h, P, bs));
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(percnn, m)
{
    m.def("pi_step(Tensor h, Tensor params, str options=\"\") -> Tensor");
    m.def("pi_step_backward(Tensor h, Tensor params, Tensor g_out, str options=\"\") -> (Tensor, Tensor)");
    m.def("pi_rollout(Tensor h0, Tensor params, SymInt steps, str options=\"\") -> Tensor");
    m.def("pi_rollout_backward(Tensor traj, Tensor params, Tensor g_traj, str options=\"\") -> (Tensor, Tensor)");
    m.class_<GradSink>("GradSink").def(torch::init<Tensor>());
    m.class_<BlockState>("BlockState").def(torch::init<Tensor>());
}

TORCH_LIBRARY_IMPL(percnn, CUDA, m)
{
    m.impl("pi_step", pi_step_impl);
    m.impl("pi_step_backward", pi_step_backward_impl);
    m.impl("pi_rollout", pi_rollout_impl);
    m.impl("pi_rollout_backward", pi_rollout_backward_impl);
}

// CPU tensors fail loudly (there is no CPU path), with the package's message instead of the dispatcher's
TORCH_LIBRARY_IMPL(percnn, CPU, m)
{
    m.impl("pi_step", pi_step_impl);
    m.impl("pi_step_backward", pi_step_backward_impl);
    m.impl("pi_rollout", pi_rollout_impl);
    m.impl("pi_rollout_backward", pi_rollout_backward_impl);
}

TORCH_LIBRARY_IMPL(percnn, Autograd, m)
{
    m.impl("pi_step", pi_step_autograd);
    m.impl("pi_rollout", pi_rollout_autograd);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "percnn_amd: eager fast path of the Pi-block step (see torch_ext.cpp)";
    // (the pack node's ctx holds the SINK, never the BlockState: BlockState -> block -> grad_fn -> ctx -> BlockState would be a
    // reference cycle through C++ that Python's collector cannot see -- ADVICE r4)
    py::class_<GradSink, c10::intrusive_ptr<GradSink>>(m, "GradSink")
        .def_property_readonly("task", [](const GradSink& s) { return s.task; });
    py::class_<BlockState, c10::intrusive_ptr<BlockState>>(m, "BlockState")
        .def_property_readonly("task", [](const BlockState& b) { return b.sink->task; })
        .def_property_readonly("sink", [](const BlockState& b) { return b.sink; })
        .def_property_readonly("holds_block", [](const BlockState& b) { return b.block.defined(); })
        .def_property_readonly("held_frames", [](const BlockState& b) { return (int64_t)b.frames.size(); })
        .def_property_readonly("spec_launches", [](const BlockState& b) { return b.spec_launches; })
        .def_property_readonly("spec_hits", [](const BlockState& b) { return b.spec_hits; })
        .def_property("speculate", [](const BlockState& b) { return b.speculate; },
                      [](BlockState& b, bool v) { std::lock_guard<std::mutex> lk(b.mu); b.speculate = v; if (!v) b.forget(); });
    m.def("abi_version", []() { return percnn_pi_abi_version(); });
    m.def("block_key", &block_key);
    m.def("block_key_src", &block_key_src);
    m.def("cell_key", [](const py::object& cell) {
        PyObject** dp = _PyObject_GetDictPtr(cell.ptr());
        return (dp && *dp) ? cell_key(*dp) : (int64_t)-1;
    });
    m.def("fast_forward", &fast_forward);
    m.def("bind_block", [](const c10::intrusive_ptr<BlockState>& bs, const Tensor& block, int64_t key) {
        std::lock_guard<std::mutex> lk(bs->mu);
        bs->block = block;
        bs->key = key;
    });
    m.def("step_nograd", &step_nograd, py::arg("h"), py::arg("params"), py::arg("state") = py::none());
    m.def("cell_step", &cell_step);
    m.def("new_block_state", [](const Tensor& like, int64_t np) {
        return c10::make_intrusive<BlockState>(at::zeros({np}, like.options().dtype(at::kDouble)));
    });
    // the cell is done with this state (a backward pass consumed the block, the block was replaced or invalidated): drop the
    // block, the speculated frames and the key, so that nothing reachable from the state refers to an autograd graph any more
    m.def("release_block", [](const c10::intrusive_ptr<BlockState>& bs) {
        std::lock_guard<std::mutex> lk(bs->mu);
        bs->forget();
        bs->block = Tensor();
        bs->key = -1;
    });
    m.def("take_block_grad", [](const c10::intrusive_ptr<GradSink>& sink, const Tensor& like) -> c10::optional<Tensor> {
        Tensor t = sink->take(current_task(), like);
        if (!t.defined()) return c10::nullopt;
        return t;
    });
}

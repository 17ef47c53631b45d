"""ctypes binding of libpercnn_pi.so (the C-ABI declared in include/percnn_pi.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised.  Build with ``python __graft_entry__.py`` or ``percnn_amd.build()``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("PERCNN_PI_LIB", os.path.join(CSRC, "libpercnn_pi.so"))
ABI_VERSION = 3

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
_INC = os.path.join("..", "..", "include")
# translation unit -> headers it depends on (all under csrc/ unless a path is given)
SOURCES = {
    "pi_abi.hip": ["pi_kernels.h", "pi_tile2d.h", "pi_stream3d.h", "pi_brick3d.h", "pi_res3d.h", "pi_adv.h", "pi_contract.h", "pi_peer.h", "pi_device.h", "pi_host.h", os.path.join(_INC, "percnn_pi.h")],
    "pi_s1_abi.hip": ["pi_s1.h", "pi_device.h", "pi_host.h", os.path.join(_INC, "percnn_pi.h"), os.path.join(_INC, "percnn_pi_stage1.h")],
    "pi_up3d_abi.hip": ["pi_up3d.h", "pi_device.h", os.path.join(_INC, "percnn_pi.h")],
}

# the PyTorch operator library + eager fast path (csrc/torch_ext.cpp: TORCH_LIBRARY(percnn) operators with C++ autograd,
# dispatcher -> C-ABI without a Python / ctypes frame); links libpercnn_pi.so, loaded by `torch_ext()` below
TORCH_EXT_SRC = os.path.join(CSRC, "torch_ext.cpp")
TORCH_EXT_PATH = os.path.join(CSRC, "percnn_torch.so")


def _torch_ext_commands():
    """(compile command, link command) of the operator library: plain g++ against the torch headers of THIS interpreter's
    PyTorch-ROCm (hip headers only for the stream type: -D__HIP_PLATFORM_AMD__ is what they ask for, not a dual path)"""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    cxx = os.environ.get("CXX", "g++")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    obj = os.path.join(CSRC, "torch_ext.o")
    inc = [f"-I{d}" for d in ce.include_paths()] + ["-I/opt/rocm/include", f"-I{sysconfig.get_paths()['include']}",
                                                    f"-I{os.path.join(CSRC, _INC)}"]
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    compile_cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
                   "-DTORCH_EXTENSION_NAME=percnn_torch", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"] + inc + ["-c", "-o", obj, TORCH_EXT_SRC]
    link_cmd = [cxx, "-shared", "-o", TORCH_EXT_PATH, obj, f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
                "-ltorch_hip", "-ltorch_python", f"-L{CSRC}", "-lpercnn_pi", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    return obj, compile_cmd, link_cmd


EXPORTS = [
    "percnn_pi_abi_version", "percnn_pi_param_count", "percnn_pi_bwd_workspace_bytes",
    "percnn_pi_rollout_bwd_workspace_bytes", "percnn_pi_set_option", "percnn_pi_persist_status", "percnn_pi_persist_fence", "percnn_pi_halo_ring_bytes",
    "percnn_pi_peer_box_bytes", "percnn_pi_peer_box_alloc", "percnn_pi_peer_box_free", "percnn_pi_peer_box_export",
    "percnn_pi_peer_box_open", "percnn_pi_peer_box_close", "percnn_pi_peer_box_status",
    "percnn_pi_peer_exchange_f32", "percnn_pi_peer_exchange_f64",
    "percnn_pi_pack_fwd_f32", "percnn_pi_pack_fwd_f64", "percnn_pi_pack_bwd_f32", "percnn_pi_pack_bwd_f64",
    "percnn_pi_pack_fwd_guard_f32", "percnn_pi_pack_fwd_guard_f64", "percnn_pi_host_words_alloc", "percnn_pi_host_words_free",
    "percnn_pi_debug_hog",
    "percnn_pi_debug_blockmap", "percnn_pi_debug_plan", "percnn_pi_residual_sqloss_workspace_bytes",
    "percnn_pi_residual_sqloss_f32", "percnn_pi_residual_sqloss_f64", "percnn_pi_residual_sqloss_bwd_f32",
    "percnn_pi_residual_sqloss_bwd_f64",
] + [f"percnn_pi_{op}_{suf}" for suf in ("f32", "f64")
     for op in ("step_fwd", "step_bwd", "rollout_fwd", "rollout_bwd", "slab_step_fwd", "slab_step_bwd", "slab_wgrad",
                "slab_step_fwd_range", "slab_step_bwd_range", "slab_rollout_fwd", "slab_rollout_bwd", "residual_fwd",
                "residual_bwd", "contract_fwd", "contract_bwd", "step_fwd_opt", "step_bwd_opt", "rollout_fwd_opt",
                "rollout_bwd_opt", "rollout_bwd_sqerr", "traj_sqerr", "step_bwd_rows", "bwd_rows_finish", "rollout_bwd_top")] + [
    "percnn_pi_s1_param_count", "percnn_pi_s1_step_fwd_f32", "percnn_pi_s1_rollout_fwd_f32",
    "percnn_pi_s1_rollout_bwd_workspace_bytes", "percnn_pi_s1_rollout_bwd_f32", "percnn_pi_s1_set_option",
    "percnn_pi_conv3d_k5c8_f32", "percnn_pi_conv3d_k5c8_wgrad_workspace_bytes", "percnn_pi_conv3d_k5c8_wgrad_f32",
]


def build(force: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile the HIP kernels + C-ABI for gfx950 with hipcc (cross-compiles without a GPU).
    One object per translation unit (rebuilt only when it or its headers changed), linked into one library."""
    out = out or os.path.join(CSRC, "libpercnn_pi.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = "" if not extra_flags else "_" + str(abs(hash(tuple(extra_flags))) % 10 ** 8)
    objs, relink = [], force or not os.path.exists(out)
    procs = []
    for src, hdrs in SOURCES.items():
        obj = os.path.join(CSRC, src.replace(".hip", tag + ".o"))
        deps = [os.path.join(CSRC, d) for d in [src] + hdrs]
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-c", "-o", obj, os.path.join(CSRC, src)]
            procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
            relink = True
    # the operator library's object compiles next to the HIP translation units (g++, ~40 s), it is linked after the library
    ext_obj, ext_compile, ext_link = _torch_ext_commands()
    ext_deps = [TORCH_EXT_SRC, os.path.join(CSRC, _INC, "percnn_pi.h")]
    ext_stale = force or not os.path.exists(ext_obj) or any(os.path.getmtime(ext_obj) < os.path.getmtime(d) for d in ext_deps)
    if ext_stale and not tag:
        procs.append((ext_compile, subprocess.Popen(ext_compile, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    relinked = relink or any(os.path.getmtime(out) < os.path.getmtime(o) for o in objs)
    if relinked:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs, cwd=CSRC)
    if not tag and out == os.path.join(CSRC, "libpercnn_pi.so") and (
            ext_stale or not os.path.exists(TORCH_EXT_PATH) or os.path.getmtime(TORCH_EXT_PATH) < os.path.getmtime(ext_obj)):
        subprocess.check_call(ext_link, cwd=CSRC)
    # what this call actually did (a tree that already carries objects / the library compiles nothing unless forced)
    global last_build
    last_build = {"build_mode": "forced" if force else "incremental",
                  "compiled": [os.path.basename(c[-1]) for c, _ in procs], "relinked": bool(relinked),
                  "library": os.path.relpath(out, os.path.dirname(_HERE))}
    return out


last_build: dict = {}


_lib = None
_torch_ext = None


def torch_ext():
    """The operator library (csrc/torch_ext.cpp), imported once: registers torch.ops.percnn.{pi_step, pi_rollout} (+ backward,
    C++ autograd) with the dispatcher and returns the module with the eager fast-path entry points.  Fails loudly when it has
    not been built or was built against another libpercnn_pi.so -- there is no Python fallback."""
    global _torch_ext
    if _torch_ext is not None:
        return _torch_ext
    if not os.path.exists(TORCH_EXT_PATH):
        raise RuntimeError(f"percnn_amd: operator library not found at {TORCH_EXT_PATH} -- build it with "
                           f"`python -c 'import percnn_amd; percnn_amd.build()'` (needs hipcc and g++).")
    lib()                                                     # libpercnn_pi.so first: same file, checked ABI
    import importlib.machinery
    import importlib.util
    loader = importlib.machinery.ExtensionFileLoader("percnn_torch", TORCH_EXT_PATH)
    spec = importlib.util.spec_from_loader("percnn_torch", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    if mod.abi_version() != ABI_VERSION:
        raise RuntimeError("percnn_amd: percnn_torch.so was linked against another ABI version of libpercnn_pi.so")
    _torch_ext = mod
    return mod


def lib() -> ctypes.CDLL:
    """Load the library once; fail loudly if it is absent or of another ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"percnn_amd: HIP library not found at {LIB_PATH}. There is no CPU fallback -- build it with "
            f"`python -c 'import percnn_amd; percnn_amd.build()'` (needs hipcc).")
    L = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError(f"percnn_amd: {LIB_PATH} does not export {name}")
    L.percnn_pi_abi_version.restype = ctypes.c_int
    if L.percnn_pi_abi_version() != ABI_VERSION:
        raise RuntimeError("percnn_amd: ABI version mismatch between python package and libpercnn_pi.so")
    vp, i64p, ci, sz = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_size_t
    L.percnn_pi_param_count.restype = sz
    L.percnn_pi_param_count.argtypes = [ci]
    L.percnn_pi_bwd_workspace_bytes.restype = sz
    L.percnn_pi_bwd_workspace_bytes.argtypes = [ci, ci, i64p, ci]
    L.percnn_pi_rollout_bwd_workspace_bytes.restype = sz
    L.percnn_pi_rollout_bwd_workspace_bytes.argtypes = [ci, ci, i64p, ci, ci]
    L.percnn_pi_set_option.restype = ci
    L.percnn_pi_set_option.argtypes = [ctypes.c_char_p, ctypes.c_long]
    L.percnn_pi_persist_status.restype, L.percnn_pi_persist_status.argtypes = ci, [ctypes.POINTER(ctypes.c_long)]
    L.percnn_pi_halo_ring_bytes.restype, L.percnn_pi_halo_ring_bytes.argtypes = sz, []
    if L.percnn_pi_halo_ring_bytes() != ctypes.sizeof(HaloRing):
        raise RuntimeError("percnn_amd: percnn_pi_halo_ring layout differs between the python binding and libpercnn_pi.so")
    vpp = ctypes.POINTER(ctypes.c_void_p)
    L.percnn_pi_peer_box_bytes.restype, L.percnn_pi_peer_box_bytes.argtypes = sz, [sz]
    L.percnn_pi_peer_box_alloc.restype, L.percnn_pi_peer_box_alloc.argtypes = ci, [vpp, sz]
    L.percnn_pi_peer_box_free.restype, L.percnn_pi_peer_box_free.argtypes = ci, [vp]
    L.percnn_pi_peer_box_export.restype, L.percnn_pi_peer_box_export.argtypes = ci, [vp, vp]
    L.percnn_pi_peer_box_open.restype, L.percnn_pi_peer_box_open.argtypes = ci, [vp, vpp]
    L.percnn_pi_peer_box_close.restype, L.percnn_pi_peer_box_close.argtypes = ci, [vp]
    L.percnn_pi_peer_box_status.restype = ci
    L.percnn_pi_peer_box_status.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), vp]
    L.percnn_pi_debug_blockmap.restype = ci
    L.percnn_pi_debug_blockmap.argtypes = [ci, i64p, ci, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.percnn_pi_debug_plan.restype = ci
    L.percnn_pi_debug_plan.argtypes = [ci, ci, i64p, ci, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    cd = ctypes.c_double
    for suf in ("f32", "f64"):
        f = getattr(L, f"percnn_pi_pack_fwd_{suf}")
        f.restype, f.argtypes = ci, [ctypes.POINTER(ParamPtrs), ci, ci, cd, cd, ci, ci, vp, vp]
        f = getattr(L, f"percnn_pi_pack_bwd_{suf}")
        f.restype, f.argtypes = ci, [ctypes.POINTER(ParamPtrs), ctypes.POINTER(ParamPtrs), ci, ci, cd, cd, ci, ci, vp, vp]
        f = getattr(L, f"percnn_pi_pack_fwd_guard_{suf}")
        f.restype, f.argtypes = ci, [ctypes.POINTER(ParamPtrs), ci, ci, cd, cd, ci, ci, vp, cd, cd, vp, cd, vp]
    L.percnn_pi_host_words_alloc.restype, L.percnn_pi_host_words_alloc.argtypes = ci, [ctypes.POINTER(ctypes.c_void_p), sz]
    L.percnn_pi_host_words_free.restype, L.percnn_pi_host_words_free.argtypes = ci, [vp]
    L.percnn_pi_debug_hog.restype, L.percnn_pi_debug_hog.argtypes = ci, [ci, ci, ci, vp]
    for suf in ("f32", "f64"):
        f = getattr(L, f"percnn_pi_peer_exchange_{suf}")
        f.restype, f.argtypes = ci, [vp, ci, i64p, ci, ci, ctypes.POINTER(PeerRing), vp]
    for suf in ("f32", "f64"):
        f = getattr(L, f"percnn_pi_contract_fwd_{suf}")
        f.restype, f.argtypes = ci, [vp, ci, vp, vp]
        f = getattr(L, f"percnn_pi_contract_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, ci, vp, vp, vp]
        f = getattr(L, f"percnn_pi_step_fwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, ci, ci, i64p, vp]
        f = getattr(L, f"percnn_pi_step_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, vp, vp, sz, vp, ci, ci, i64p, vp]
        f = getattr(L, f"percnn_pi_slab_step_fwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, ci, ci, i64p, ci, ci, vp]
        f = getattr(L, f"percnn_pi_slab_step_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, vp, vp, sz, vp, ci, ci, i64p, ci, ci, vp]
        f = getattr(L, f"percnn_pi_slab_step_fwd_range_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, ci, ci, i64p, ci, ci, ci, vp]
        f = getattr(L, f"percnn_pi_slab_step_bwd_range_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, vp, vp, sz, vp, ci, ci, i64p, ci, ci, ci, ci, vp]
        f = getattr(L, f"percnn_pi_slab_rollout_fwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, ci, ci, i64p, ci, ci, vp, ci, vp]
        f = getattr(L, f"percnn_pi_slab_rollout_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, vp, sz, vp, ci, ci, i64p, ci, ci, vp, ci, vp]
        f = getattr(L, f"percnn_pi_slab_wgrad_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, sz, vp, ci, ci, i64p, ci, ci, vp]
        f = getattr(L, f"percnn_pi_residual_fwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, ci, i64p, ci, vp]
        f = getattr(L, f"percnn_pi_residual_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, ci, i64p, ci, vp]
        f = getattr(L, f"percnn_pi_residual_sqloss_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, ci, i64p, ci, ci, vp, vp, sz, vp]
        f = getattr(L, f"percnn_pi_residual_sqloss_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, ci, i64p, ci, ci, ci, vp, vp, vp]
        f = getattr(L, f"percnn_pi_rollout_fwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, ci, ci, i64p, ci, vp]
        f = getattr(L, f"percnn_pi_rollout_bwd_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, ctypes.c_char_p, vp, vp, vp, sz, vp, ci, ci, i64p, ci, vp]
        cs = ctypes.c_char_p
        f = getattr(L, f"percnn_pi_step_fwd_opt_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, ci, ci, i64p, cs, vp]
        f = getattr(L, f"percnn_pi_step_bwd_opt_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, vp, vp, vp, vp, sz, vp, ci, ci, i64p, cs, vp]
        f = getattr(L, f"percnn_pi_rollout_fwd_opt_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, ci, ci, i64p, ci, cs, vp]
        f = getattr(L, f"percnn_pi_rollout_bwd_opt_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, cs, vp, vp, vp, sz, vp, ci, ci, i64p, ci, cs, vp]
        f = getattr(L, f"percnn_pi_rollout_bwd_sqerr_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, cs, cd, vp, vp, vp, vp, sz, vp, ci, ci, i64p, ci, cs, vp]
        f = getattr(L, f"percnn_pi_traj_sqerr_{suf}")
        f.restype, f.argtypes = ci, [vp, vp, cs, ci, ci, i64p, cd, vp, vp, sz, vp]
    L.percnn_pi_residual_sqloss_workspace_bytes.restype, L.percnn_pi_residual_sqloss_workspace_bytes.argtypes = sz, []
    L.percnn_pi_s1_param_count.restype = sz
    L.percnn_pi_s1_param_count.argtypes = []
    L.percnn_pi_s1_step_fwd_f32.restype, L.percnn_pi_s1_step_fwd_f32.argtypes = ci, [vp, vp, vp, i64p, vp]
    L.percnn_pi_s1_rollout_fwd_f32.restype, L.percnn_pi_s1_rollout_fwd_f32.argtypes = ci, [vp, vp, i64p, ci, vp]
    L.percnn_pi_s1_rollout_bwd_workspace_bytes.restype = sz
    L.percnn_pi_s1_rollout_bwd_workspace_bytes.argtypes = [i64p, ci]
    L.percnn_pi_conv3d_k5c8_f32.restype, L.percnn_pi_conv3d_k5c8_f32.argtypes = ci, [vp, vp, vp, vp, i64p, vp]
    L.percnn_pi_conv3d_k5c8_wgrad_workspace_bytes.restype, L.percnn_pi_conv3d_k5c8_wgrad_workspace_bytes.argtypes = sz, []
    L.percnn_pi_conv3d_k5c8_wgrad_f32.restype = ci
    L.percnn_pi_conv3d_k5c8_wgrad_f32.argtypes = [vp, vp, vp, vp, sz, i64p, vp]
    L.percnn_pi_s1_set_option.restype, L.percnn_pi_s1_set_option.argtypes = ci, [ctypes.c_char_p, ctypes.c_long]
    L.percnn_pi_s1_rollout_bwd_f32.restype = ci
    L.percnn_pi_s1_rollout_bwd_f32.argtypes = [vp, vp, ctypes.c_char_p, vp, vp, vp, sz, vp, i64p, ci, vp]
    # process-wide tuning defaults from the environment ("key=value,..."; the keys of percnn_pi_set_option) -- what a launcher
    # uses to A/B an option in worker processes it does not construct itself
    spec = os.environ.get("PERCNN_PI_OPTIONS", "")
    for kv in filter(None, spec.split(",")):
        k, _, v = kv.partition("=")
        if L.percnn_pi_set_option(k.strip().encode(), int(v)) != 0:
            raise RuntimeError(f"percnn_amd: PERCNN_PI_OPTIONS: bad option {kv!r}")
    _lib = L
    return L


class ParamPtrs(ctypes.Structure):
    """percnn_pi_param_ptrs of include/percnn_pi.h: device pointers of the reference's parameter tensors"""
    _fields_ = [("c", ctypes.c_void_p * 2), ("w", ctypes.c_void_p), ("branch", ctypes.c_void_p * 16)]


class PeerRing(ctypes.Structure):
    """percnn_pi_peer_ring of include/percnn_pi.h: this rank's mailbox, the mapped mailboxes of its ring neighbours"""
    _fields_ = [("my_box", ctypes.c_void_p), ("prev_box", ctypes.c_void_p), ("next_box", ctypes.c_void_p),
                ("slot_bytes", ctypes.c_size_t), ("epoch", ctypes.c_uint64), ("timeout_ticks", ctypes.c_uint64)]


class HaloRing(ctypes.Structure):
    """percnn_pi_halo_ring of include/percnn_pi.h: communicator, neighbours and the four RCCL entry points as addresses,
    or (``peer`` set) the peer-mailbox transport"""
    _fields_ = [("comm", ctypes.c_void_p), ("prev", ctypes.c_int), ("next", ctypes.c_int),
                ("dtype_f32", ctypes.c_int), ("dtype_f64", ctypes.c_int),
                ("group_start", ctypes.c_void_p), ("group_end", ctypes.c_void_p),
                ("send", ctypes.c_void_p), ("recv", ctypes.c_void_p), ("peer", ctypes.POINTER(PeerRing)),
                ("stage", ctypes.c_void_p), ("stage_bytes", ctypes.c_size_t)]


def persist_fence(stream=None) -> None:
    """percnn_pi_persist_fence: wait for `stream` (default: torch's current stream) and raise if a fire-and-forget resident launch
    (persist_handshake = 0) aborted since anybody looked -- for callers that hand rollout outputs to code outside this package."""
    import torch
    st = torch.cuda.current_stream() if stream is None else stream
    check(lib().percnn_pi_persist_fence(ctypes.c_void_p(st.cuda_stream)), "persist_fence")


def persist_status() -> dict:
    """percnn_pi_persist_status: what the persistent tile sweep has done in this process (launches, aborts, whether the current
    device fell back to one launch per group of steps, where the last abort happened)."""
    info = (ctypes.c_long * 8)()
    lib().percnn_pi_persist_status(info)
    return {"launches": info[0], "aborts": info[1], "disabled_on_current_device": bool(info[2]), "last_abort_group": info[3],
            "last_abort_tile": info[4], "last_launch_state": info[5]}


def _persist_note() -> str:
    try:
        return f"persist_status() = {persist_status()}"
    except Exception:                                        # pragma: no cover
        return ""


class GridTooLargeError(RuntimeError):
    """PERCNN_PI_ETOOLARGE: nothing was launched; callers with another route for such grids catch this."""


def check(rc: int, what: str) -> None:
    if rc == -3:
        raise GridTooLargeError(f"percnn_amd: {what} failed: grid too large for this kernel family's 32-bit offsets / fixed "
                                f"workspace (a 2D field or a 3D plane of >= 4 GiB per species, or more workgroups than the "
                                f"loss pass has partial slots)")
    if rc != 0:
        names = {-1: "invalid argument", -2: "workspace too small",
                 -3: "grid too large for the step kernels' 32-bit offsets (a 2D field or a 3D plane of >= 4 GiB per species)",
                 -4: "an EARLIER call's persistent launch (resident forward or tile sweep) aborted on the device (its workgroups "
                     "could not all be resident: another process / kernel holds CUs, or a CU mask is set) -- the results of "
                     "that earlier persistent launch (forward or backward) are "
                     "invalid; this call launched nothing.  " + _persist_note()}
        raise RuntimeError(f"percnn_amd: {what} failed: {names.get(rc, 'hipError_t ' + str(rc))}")


_shape_args: dict = {}


def shape_arg(shape):
    """int64 array of a grid shape for the C-ABI (cached per shape: the per-step entry points of a reference-style loop
    build the same few arrays thousands of times; the arrays are never written to)"""
    key = tuple(int(s) for s in shape)
    a = _shape_args.get(key)
    if a is None:
        if len(_shape_args) > 256:
            _shape_args.clear()
        a = _shape_args[key] = (ctypes.c_int64 * len(key))(*key)
    return a


FAMILIES = {0: "direct", 1: "tile2d", 2: "stream3d", 3: "brick3d", 4: "advective"}


def rollout_plan(hc: int, shape, elem_size: int, options=None) -> dict:
    """Kernel families a rollout of this problem runs on (the library's own dispatch, ``percnn_pi_debug_plan``)."""
    out = (ctypes.c_int * 15)()
    check(lib().percnn_pi_debug_plan(int(hc), len(shape), shape_arg(shape), int(elem_size), options_arg(options), out),
          "debug_plan")
    return {"fwd": FAMILIES[out[0]], "bwd": FAMILIES[out[1]], "fused_gradients": bool(out[2]), "fwd_steps_per_launch": out[3],
            "bwd_steps_per_launch": out[4], "fwd_planes_per_pass": out[5], "bwd_planes_per_pass": out[6],
            "brick_lanes": out[7], "tile": (out[8], out[9], out[10]) if out[10] else None,
            "tile_fwd": (out[11], out[12], out[13]) if out[13] else None, "bwd_persistent": bool(out[14] & 1),
            "fwd_persistent": bool(out[14] & 2)}


def set_option(key: str, value: int) -> None:
    """Process-wide DEFAULT of a tuning option (include/percnn_pi.h); per-call overrides: the ``options`` argument of
    ``functional.rollout_fwd_/rollout_bwd/step_fwd/step_bwd`` and of the ``torch.ops.percnn`` operators."""
    check(lib().percnn_pi_set_option(key.encode(), int(value)), f"set_option({key}={value})")


def options_arg(options) -> bytes | None:
    """dict / "k=v,k=v" string / None -> the `options` C string of the *_opt entry points"""
    if not options:
        return None
    if isinstance(options, str):
        return options.encode()
    return ",".join(f"{k}={int(v)}" for k, v in options.items()).encode()

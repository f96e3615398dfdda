"""Hand-scheduled forward / backward of the ActorCritic for the MI355X (no reference analogue: the reference lets
autograd run go1_gym_learn/ppo_cse/ppo.py:99-205 as ~350 small kernels per mini-batch).

Same mathematics as `FlatPolicy.forward` + autograd, organised as
  * hipBLASLt GEMMs (torch.mm / torch.addmm with `out=`) on buffers allocated once per batch size,
  * the fused element-wise / reduction / small-wgrad kernels of csrc/go1ppo.hip (C-ABI include/go1ppo.h) in between,
  * parameter gradients written straight into the fp32 flat gradient the optimiser and the RCCL all-reduce use.
The whole thing is capture-safe (static addresses, no host reads), so `PPO._capture` records it as a HIP graph.

bf16 compute only (BASELINE config 2's "bf16 policy"); the fp32 / CPU path stays on autograd (`ppo.py`), which is
also the checker for this one (tests/test_gpu_env.py).
"""
import ctypes
import os

import torch

HEAD = 64
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libgo1ppo.so"))
_lib = None


class Go1PpoLibraryMissing(RuntimeError):
    pass


class LossArgs(ctypes.Structure):
    _fields_ = [("mean", ctypes.c_void_p), ("value", ctypes.c_void_p), ("std", ctypes.c_void_p),
                ("head_ld", ctypes.c_int32), ("num_actions", ctypes.c_int32), ("rows", ctypes.c_int64),
                ("idx", ctypes.c_void_p), ("actions", ctypes.c_void_p), ("old_mu", ctypes.c_void_p),
                ("old_sigma", ctypes.c_void_p), ("old_logp", ctypes.c_void_p), ("advantages", ctypes.c_void_p),
                ("returns", ctypes.c_void_p), ("old_values", ctypes.c_void_p),
                ("clip_param", ctypes.c_float), ("value_loss_coef", ctypes.c_float), ("entropy_coef", ctypes.c_float),
                ("use_clipped_value_loss", ctypes.c_int32),
                ("d_mean", ctypes.c_void_p), ("d_value", ctypes.c_void_p), ("d_std", ctypes.c_void_p),
                ("d_mean_bias", ctypes.c_void_p), ("d_value_bias", ctypes.c_void_p), ("kl", ctypes.c_void_p),
                ("value_loss", ctypes.c_void_p), ("surrogate_loss", ctypes.c_void_p)]


class WgradProblem(ctypes.Structure):
    _fields_ = [("dz", ctypes.c_void_p), ("h", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("bias_grad", ctypes.c_void_p),
                ("rows", ctypes.c_int64), ("ld_dz", ctypes.c_int32), ("ld_h", ctypes.c_int32), ("n", ctypes.c_int32),
                ("k", ctypes.c_int32), ("ldw", ctypes.c_int32), ("chunk_rows", ctypes.c_int32), ("wg_offset", ctypes.c_int32),
                ("zero_n", ctypes.c_int32), ("zero_k0", ctypes.c_int32), ("zero_k1", ctypes.c_int32),
                ("partials", ctypes.c_void_p), ("partial_stride", ctypes.c_int64)]


class GradPiece(ctypes.Structure):
    _fields_ = [("begin", ctypes.c_int64), ("count", ctypes.c_int64), ("src", ctypes.c_void_p), ("stride", ctypes.c_int64),
                ("kind", ctypes.c_int32), ("slabs", ctypes.c_int32), ("cols", ctypes.c_int32), ("zero_rows", ctypes.c_int32),
                ("zero_c0", ctypes.c_int32), ("zero_c1", ctypes.c_int32)]


class TailLayer(ctypes.Structure):
    _fields_ = [("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n_out", ctypes.c_int32),
                ("k_in", ctypes.c_int32), ("ld_out", ctypes.c_int32), ("elu", ctypes.c_int32)]


class TailNet(ctypes.Structure):
    _fields_ = [("in_", ctypes.c_void_p), ("rows", ctypes.c_int64), ("ld_in", ctypes.c_int32), ("num_layers", ctypes.c_int32),
                ("elu_in", ctypes.c_int32), ("npv", ctypes.c_int32), ("latent", ctypes.c_void_p), ("wz", ctypes.c_void_p),
                ("lat_ld", ctypes.c_int32), ("wz_ld", ctypes.c_int32),
                ("layer", TailLayer * 4)]


class TailArgs(ctypes.Structure):
    _fields_ = [("num_nets", ctypes.c_int32), ("_pad", ctypes.c_int32), ("net", TailNet * 3)]


class Mlp2Fwd(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "W2", "b2", "W3", "b3", "z2", "out")] + [("rows", ctypes.c_int64)] + \
               [(n, ctypes.c_int32) for n in ("ld_x", "ld_z2", "ld_out", "elu_input")]


class Mlp2Bwd(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("d_out", "z2", "h", "W2", "W3", "d_z2", "d_x")] + [("rows", ctypes.c_int64)] + \
               [(n, ctypes.c_int32) for n in ("ld_dout", "ld_z2", "ld_h", "ld_dz2", "ld_dx", "_pad")]


MLP2_DIMS = (256, 128, 64)


class GemmArgs(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("H", ctypes.c_void_p)] + [(n, ctypes.c_int32) for n in
                                           ("M", "N", "K", "lda", "ldb", "ldc", "ldh", "epilogue", "elu_c0", "elu_c1", "elu_skip_c0", "elu_skip_c1", "bias_bf16", "_pad")]


class _AdamTranspose(ctypes.Structure):
    _fields_ = [("start", ctypes.c_int64), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("dst", ctypes.c_void_p)]


class AdamExtras(ctypes.Structure):
    _fields_ = [("num_transposes", ctypes.c_int32), ("_pad", ctypes.c_int32), ("transpose", _AdamTranspose * 2)]


EXPORTED_SYMBOLS = ["go1ppo_mlp2_fwd", "go1ppo_mlp2_bwd", "go1ppo_gemm_nt", "go1ppo_gemm_nt_pair", "go1ppo_sum_partials", "go1ppo_wgrad_tn_plan", "go1ppo_wgrad_tn_batched", "go1ppo_tail_fwd", "go1ppo_elu_fwd", "go1ppo_elu_bwd", "go1ppo_loss", "go1ppo_mse", "go1ppo_wgrad", "go1ppo_wgrad_plan",
                    "go1ppo_wgrad_batched", "go1ppo_act",
                    "go1ppo_store_step", "go1ppo_ring_snapshot", "go1ppo_ring_step", "go1ppo_ring_gather", "go1ppo_gae", "go1ppo_normalize", "go1ppo_opt_partials", "go1ppo_opt_prestep",
                    "go1ppo_opt_prestep_pieces", "go1ppo_grad_reduce", "go1ppo_opt_adam", "go1ppo_version"]


def load_library(path=None):
    """dlopen libgo1ppo.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise Go1PpoLibraryMissing(f"{p} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(hipcc --offload-arch=gfx950); the fused PPO update has no CPU fallback")
    L = ctypes.CDLL(p)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.go1ppo_elu_fwd.argtypes = [vp, i64, i32, i32, vp, i32, i32, vp, i32, i32, vp]
    L.go1ppo_elu_bwd.argtypes = [vp, i32, vp, i32, i64, i32, vp, vp, i32, vp]
    L.go1ppo_loss.argtypes = [ctypes.POINTER(LossArgs), vp]
    L.go1ppo_mse.argtypes = [vp, i32, vp, i32, vp, i64, i64, i32, vp, vp, vp, vp, vp]
    L.go1ppo_wgrad.argtypes = [vp, i32, vp, i32, i64, i32, i32, vp, i32, vp, vp]
    L.go1ppo_wgrad_plan.argtypes = [ctypes.POINTER(WgradProblem), i32]
    L.go1ppo_tail_fwd.argtypes = [ctypes.POINTER(TailArgs), vp]
    L.go1ppo_gemm_nt.argtypes = [ctypes.POINTER(GemmArgs), vp]
    L.go1ppo_gemm_nt_pair.argtypes = [ctypes.POINTER(GemmArgs), ctypes.POINTER(GemmArgs), vp]
    L.go1ppo_sum_partials.argtypes = [vp, i32, i64, i64, i32, vp, i32, i32, i32, vp]
    L.go1ppo_mlp2_fwd.argtypes = [ctypes.POINTER(Mlp2Fwd), i32, vp]
    L.go1ppo_mlp2_bwd.argtypes = [ctypes.POINTER(Mlp2Bwd), i32, vp]
    L.go1ppo_wgrad_batched.argtypes = [vp, i32, i32, vp]
    L.go1ppo_wgrad_tn_plan.argtypes = [ctypes.POINTER(WgradProblem), i32]
    L.go1ppo_wgrad_tn_batched.argtypes = [vp, i32, i32, vp]
    f32 = ctypes.c_float
    L.go1ppo_act.argtypes = [vp, vp, i32, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp]
    L.go1ppo_store_step.argtypes = [vp, vp, vp, vp, vp, f32, i64, vp, vp, vp, vp]
    L.go1ppo_ring_snapshot.argtypes = [vp, i64, i64, i32, i32, vp, vp]
    L.go1ppo_ring_step.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp]
    L.go1ppo_ring_gather.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, vp, vp]
    L.go1ppo_gae.argtypes = [vp, vp, vp, vp, i32, i64, f32, f32, vp, vp, vp, vp]
    L.go1ppo_normalize.argtypes = [vp, i64, vp, vp]
    L.go1ppo_opt_partials.argtypes = []
    L.go1ppo_opt_prestep.argtypes = [vp, i64, f32, vp, vp, vp, vp, f32, f32, f32, f32, vp]
    L.go1ppo_opt_prestep_pieces.argtypes = [vp, i64, vp, i32, f32, vp, vp, vp, vp, f32, f32, f32, f32, vp]
    L.go1ppo_grad_reduce.argtypes = [vp, vp, i32, vp]
    L.go1ppo_opt_adam.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, f32, vp, f32, vp, vp, f32, f32, f32, vp, i64, vp, i64, i32, vp,
                                  ctypes.POINTER(AdamExtras), vp]
    for name in EXPORTED_SYMBOLS[:-1]:
        getattr(L, name).restype = ctypes.c_int
    L.go1ppo_version.restype = ctypes.c_char_p
    if path is None:
        _lib = L
    return L


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


def gemm_args(a, b, c, bias=None, elu=None, elu_bwd_of=None, elu_skip=None):
    """Go1PpoGemmArgs for c = epilogue(a @ b.T + bias): `elu` = (c0, c1) column range to activate (True: all columns) except the
    columns `elu_skip` = (s0, s1); `elu_bwd_of` = post-ELU activations H, the result is multiplied by elu'(H)."""
    g = GemmArgs()
    g.A, g.B, g.C, g.bias, g.H = a.data_ptr(), b.data_ptr(), c.data_ptr(), _ptr(bias), _ptr(elu_bwd_of)
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = a.shape[0], b.shape[0], a.shape[1], _ld(a), _ld(b), _ld(c)
    assert b.shape[1] == g.K and tuple(c.shape) == (g.M, g.N) and (elu is None or elu_bwd_of is None)
    assert a.dtype == b.dtype == c.dtype == torch.bfloat16 and (bias is None or bias.dtype in (torch.float32, torch.bfloat16))
    g.bias_bf16 = int(bias is not None and bias.dtype == torch.bfloat16)
    if elu_bwd_of is not None:
        assert elu_bwd_of.shape == c.shape and elu_bwd_of.dtype == torch.bfloat16
        g.epilogue, g.ldh = 2, _ld(elu_bwd_of)
    elif elu is not None:
        g.epilogue = 1
        g.elu_c0, g.elu_c1 = (0, g.N) if elu is True else elu
        if elu_skip is not None:
            g.elu_skip_c0, g.elu_skip_c1 = elu_skip
    return g


def gemm_nt(lib, a, b, c, bias=None, elu=None, elu_bwd_of=None):
    g = gemm_args(a, b, c, bias, elu, elu_bwd_of)
    _chk(lib.go1ppo_gemm_nt(ctypes.byref(g), _stream()), "go1ppo_gemm_nt")
    return c


def gemm_nt_pair(lib, first, second):
    """two go1ppo_gemm_nt problems with the same tile grid and epilogue in one launch; first / second: keyword dicts of gemm_args"""
    ga, gb = gemm_args(**first), gemm_args(**second)
    _chk(lib.go1ppo_gemm_nt_pair(ctypes.byref(ga), ctypes.byref(gb), _stream()), "go1ppo_gemm_nt_pair")


class FusedNet:
    """Static-buffer forward (and optionally backward) of the three MLPs for a fixed row count M.

    `body` is the bf16 flat parameter buffer (FlatPolicy layout), `grad` the fp32 flat gradient (or None for
    inference-only engines)."""

    def __init__(self, policy, body, grad, M, lib, with_grad, two_streams=True, fused_tails=None):
        assert body.dtype == torch.bfloat16 and body.is_cuda
        self.pol, self.M, self.lib = policy, M, lib
        dev = body.device
        self.P = {name: policy._block(body, name) for name, _ in policy.blocks}
        self.G = {name: policy._block(grad, name) for name, _ in policy.blocks} if grad is not None else None
        self.nd, self.na, self.nc = policy.first
        self.n1 = self.nd + self.na + self.nc
        self.depth = {n: len(l) for n, l in policy.nets.items()}
        bf = dict(device=dev, dtype=torch.bfloat16)
        self.Y1 = torch.zeros(M, self.n1, **bf)
        self.Y1d = torch.zeros(M, self.nd, **bf)                # adaptation-only pass
        self.Z = {n: {li: torch.zeros(M, self.P[f"{n}.{li}.W"].shape[0], **bf) for li in range(1, d)}
                  for n, d in self.depth.items()}
        assert policy.npv <= HEAD and self.P["Wz"].shape == (self.na, HEAD)
        self._recording, self._batched, self._plans = None, False, {}
        # weight gradients as per-row-chunk SLABS (plain stores) summed in a fixed order by one pass — the optimiser's norm pass when
        # `defer_grad_sum` is set by the owner (FusedAdam.step_(pieces=...)), go1ppo_grad_reduce at the end of the backward pass
        # otherwise — instead of fp32 atomics into the flat gradient (15.6 of the 64.5 us of the PPO pass's batched launch on MI355X,
        # profiles/r04_wgrad_atomics_probe.txt; and run-to-run reproducible).  GO1_WGRAD_SLABS=0: the atomics.
        self._grad = grad
        self._slabs = grad is not None and os.environ.get("GO1_WGRAD_SLABS", "1") == "1"
        self.defer_grad_sum = False
        self._rec_pieces, self._slab_ws, self._last_pieces = None, {}, {}
        # weight gradients: 128-tile kernel for the batched tails; first layer (PPO pass, adaptation pass) on it or on
        # hipBLASLt.  GO1_WGRAD = "<tails><ppo W1><adaptation W1>" digits for A/B runs (tools/), default below.
        knob = os.environ.get("GO1_WGRAD", "101")
        self._wgrad_tn = knob[0] == "1" and M % 64 == 0
        self._w1_tn = (self._wgrad_tn and knob[1] == "1", self._wgrad_tn and knob[2] == "1")
        self._tails = self._tails_ad = None
        # measured on MI355X: one fused launch per dependency level beats 8 GEMMs + 5 ELU kernels 3.5x at M = 4096 (32 vs 110 us) — but the
        # engine below (ELU pass, paired 512 -> 256 GEMM, LDS-resident MLP ends) beats it at the rollout's 4096 rows since round 4
        # (rollout 6.8 vs 7.0 ms per iteration, same-box alternating runs) and at the update's 24576: the fused kernel keeps the small batches
        if fused_tails is None:
            fused_tails = M <= int(os.environ.get("GO1_FUSED_TAILS_MAX_ROWS", "2048"))     # (tests lower it to reach the other engine)
        # inference-only engines: the first layer's ELU (+ the actor's latent columns) is applied by the tail kernel while it
        # stages its input rows — two element-wise launches per rollout step less; an engine with a backward pass keeps the
        # separate pass, which leaves the activated first layer in memory for the weight gradients
        self._elu_on_load = not with_grad and os.environ.get("GO1_ELU_ON_LOAD", "1") == "1"
        if fused_tails and self._fused_tails_ok():
            nd, na = self.nd, self.na
            eol = self._elu_on_load
            self._tails = (self._tail_args([("adaptation", self.Y1[:, :nd])], with_grad, elu_in=eol),
                           self._tail_args([("actor", self.Y1[:, nd:nd + na]), ("critic", self.Y1[:, nd + na:])], with_grad, elu_in=eol,
                                           latent_for="actor" if eol else None))
            self._tails_ad = self._tail_args([("adaptation", self.Y1d)], with_grad)
        # large batches: the last two layers of every net (256 -> 128 -> 64) on the LDS-resident kernels of
        # csrc/go1ppo_mlp.h, forward and backward (GO1_MLP2=0 switches back to per-layer GEMM + ELU launches)
        self._mlp2 = (self._tails is None and os.environ.get("GO1_MLP2", "1") == "1" and self.nd == MLP2_DIMS[0] and
                      all(self.depth[n] >= 3 and tuple(self.P[f"{n}.{self.depth[n] - 2}.W"].shape) == (MLP2_DIMS[1], MLP2_DIMS[0])
                          and tuple(self.P[f"{n}.{self.depth[n] - 1}.W"].shape) == (MLP2_DIMS[2], MLP2_DIMS[1]) for n in self.depth)
                      and self.depth["adaptation"] == 3 and self.depth["actor"] == 4 and self.depth["critic"] == 4)
        self._mlp2_cache = {}
        # (the first-layer forward stays on hipBLASLt: 96 us = 1.39 PFLOP/s in situ.  A 256-tile LDS-DMA GEMM of this repository with the ELU in
        #  its epilogue measured 135-140 us — bound by the L2 -> LDS staging rate of ~10.5 TB/s chip-wide, tools/probes/gemm256_probe.hip,
        #  DESIGN.md section 7 — and was withdrawn)
        # go1ppo_gemm_nt wants 64-column tiles and 16-byte aligned rows / biases, the paired launch equal tile grids for its two problems:
        # other widths (an actor / critic first hidden width that is not a multiple of 64, or that differ) take torch.addmm / two launches
        na1, nc1 = int(self.P["actor.1.W"].shape[0]), int(self.P["critic.1.W"].shape[0])
        nt_ok = all(v % 64 == 0 for v in (na1, nc1, self.na, self.n1 - self.nd - self.na)) and \
            all(self.P[k].data_ptr() % 16 == 0 for k in ("actor.1.b", "critic.1.b", "actor.1.W", "critic.1.W"))
        pair_ok = nt_ok and na1 == nc1 and self.na == self.n1 - self.nd - self.na
        self._l1_nt = os.environ.get("GO1_L1_NT", "1") == "1" and nt_ok
        # the actor's and the critic's 512 -> 256 GEMMs (forward, input gradient) as one launch each instead of two launches on two
        # streams: tools/timeline.py showed 5-10 us of idle device at every graph fork and join (GO1_GEMM_PAIR=0: the two streams)
        self._pair = os.environ.get("GO1_GEMM_PAIR", "1") == "1" and pair_ok
        # the critic's tail is independent of the actor / adaptation chain: it runs on a side stream (forked from and
        # joined back into the caller's stream, so HIP-graph capture records it as a parallel branch)
        self._side = torch.cuda.Stream(device=dev) if two_streams else None
        if with_grad:
            self.X = torch.zeros(M, policy.Kp, **bf)
            # GEMM outputs are never strided views: the tails' input gradients land in one contiguous buffer per net
            # (dH1), the ELU-backward kernel writes them into the column blocks of dY1 for the single W1 wgrad GEMM
            self.dH1 = {"adaptation": torch.zeros(M, self.nd, **bf), "actor": torch.zeros(M, self.na, **bf),
                        "critic": torch.zeros(M, self.nc, **bf)}
            self.dY1 = torch.zeros(M, self.n1, **bf)
            self.dZ = {n: {li: torch.zeros_like(z) for li, z in zs.items()} for n, zs in self.Z.items()}
            # first-layer weight gradient on hipBLASLt: the output is small (n1 x Kp) and the reduction long (M rows), which
            # the library's single-GEMM kernels split badly (0.74 PFLOP/s); as a BATCHED GEMM over row chunks — a manual
            # split-K, partial products in bf16, summed into the fp32 gradient by the pass that used to be the cast — it
            # runs 25 % faster (tools/probes/wgrad_splitk.py: 4 chunks of 6144 rows at M = 24576)
            self._w1_split = next((b for b in (M // 6144, 4, 2) if b >= 2 and M % b == 0 and (M // b) % 64 == 0), 1)
            if os.environ.get("GO1_W1_SPLIT"):
                self._w1_split = int(os.environ["GO1_W1_SPLIT"])
            self._w1_tmp = torch.zeros(self._w1_split, self.n1, policy.Kp, **bf)
            # GO1_DGRAD_NT: the 512 -> 256 input gradients on go1ppo_gemm_nt with the ELU' epilogue instead of hipBLASLt + the
            # element-wise pass.  In situ A/B on one box: 23.30 vs 23.43 ms per iteration (it was 0.6 ms SLOWER while the
            # epilogue still loaded its ELU' operand where it used it: 19 exposed HBM round trips per workgroup)
            self._dgrad_nt = os.environ.get("GO1_DGRAD_NT", "1") == "1" and self._mlp2 and nt_ok
            self._WT = {n: torch.zeros(self.P[f"{n}.1.W"].shape[1], self.P[f"{n}.1.W"].shape[0], **bf) for n in ("actor", "critic")} \
                if self._dgrad_nt else None
            # the K-contiguous copies are refreshed per backward pass (two transpose-copy launches) until an optimiser takes them
            # over (`adam_transposes()` -> FusedAdam.set_transposes: the step that changes the weights rewrites the copies)
            self._wt_by_optimizer = False
            # privileged-observation columns of the adaptation module's and the actor's first-layer rows: structurally zero weights
            self.priv_mask = (self.nd + self.na, policy.K + 1, policy.K + 1 + policy.npv)

    # ---- kernels -----------------------------------------------------------------------------------------
    def _elu(self, y, lat=None, lat_cols=0):
        wz = self.P["Wz"] if lat is not None else None
        _chk(self.lib.go1ppo_elu_fwd(y.data_ptr(), y.shape[0], y.shape[1], _ld(y), _ptr(lat), _ld(lat) if lat is not None else 0,
                                     self.pol.npv if lat is not None else 0, _ptr(wz), HEAD, lat_cols, _stream()), "go1ppo_elu_fwd")

    def _elu_bwd(self, d, h, bias_grad, out=None):
        out = d if out is None else out
        _chk(self.lib.go1ppo_elu_bwd(d.data_ptr(), _ld(d), _ptr(h), _ld(h) if h is not None else 0, d.shape[0], d.shape[1],
                                     _ptr(bias_grad), out.data_ptr(), _ld(out), _stream()), "go1ppo_elu_bwd")

    def _wgrad(self, dz, h, gW, gb=None, zero=None):
        """weight (+ bias) gradient of one layer.  All operands are static buffers, so the calls of a backward pass are
        recorded once (`_plan`) and afterwards executed together as ONE batched launch at the end of the pass.
        zero = (rows, c0, c1): structural zeros of gW (Go1PpoWgradProblem) — batched launches only."""
        n, k = dz.shape[1], h.shape[1]
        assert gW.shape == (n, k) and gW.is_contiguous()
        if self._recording is not None:
            self._recording.append((dz, h, gW, gb, zero))
            return
        if self._batched:
            return
        assert zero is None or zero[0] == 0, "the structural zeros are applied by the batched launch"
        _chk(self.lib.go1ppo_wgrad(dz.data_ptr(), _ld(dz), h.data_ptr(), _ld(h), dz.shape[0], n, k, gW.data_ptr(), k, _ptr(gb),
                                   _stream()), "go1ppo_wgrad")

    # ---- forward -------------------------------------------------------------------------------------------
    def _tail_args(self, nets_and_inputs, keep_intermediates, elu_in=False, latent_for=None):
        """Go1PpoTailArgs for go1ppo_tail_fwd: the listed nets' layers behind the first one, reading their first-layer block
        (post-ELU, or with elu_in the pre-activation, activated on the way in — `latent_for`: the net whose input also gets
        the latent columns latent Wz^T) and writing Z[net][li] (intermediates only when the backward pass needs them)."""
        a = TailArgs()
        a.num_nets = len(nets_and_inputs)
        for N, (net, h) in zip(a.net, nets_and_inputs):
            d = self.depth[net]
            N.in_, N.rows, N.ld_in, N.num_layers = h.data_ptr(), h.shape[0], _ld(h), d - 1
            N.elu_in = int(bool(elu_in))
            if elu_in and latent_for == net:
                lat, wz = self.Z["adaptation"][self.depth["adaptation"] - 1], self.P["Wz"]
                N.latent, N.wz, N.lat_ld, N.wz_ld, N.npv = lat.data_ptr(), wz.data_ptr(), _ld(lat), HEAD, self.pol.npv
            for li in range(1, d):
                L, W, z = N.layer[li - 1], self.P[f"{net}.{li}.W"], self.Z[net][li]
                last = li == d - 1
                L.W, L.bias = W.data_ptr(), self.P[f"{net}.{li}.b"].data_ptr()
                L.out = z.data_ptr() if (last or keep_intermediates) else None
                L.n_out, L.k_in, L.ld_out, L.elu = W.shape[0], W.shape[1], _ld(z), 0 if last else 1
        return a

    def _fused_tails_ok(self):
        ok = all(self.depth[n] - 1 <= 4 for n in self.depth)
        for n, d in self.depth.items():
            for li in range(1, d):
                r, k = self.P[f"{n}.{li}.W"].shape
                ok = ok and k % 32 == 0 and k <= 512 and r % 16 == 0 and r <= 512
        return ok

    def _tail(self, net, h):
        P, d = self.P, self.depth[net]
        for li in range(1, d):
            z = self.Z[net][li]
            torch.addmm(P[f"{net}.{li}.b"], h, P[f"{net}.{li}.W"].t(), out=z)
            if li < d - 1:
                self._elu(z)
            h = z
        return h

    def forward(self, x):
        """x: (M, Kp) augmented rows.  Returns (mean (M, HEAD), value (M, HEAD), latent (M, HEAD)) padded head
        outputs (valid columns: num_actions / 1 / num_privileged_obs); views of static buffers."""
        nd, na = self.nd, self.na
        if self._mlp2:
            return self._forward_mlp2(x)
        torch.mm(x, self.P["W1"].t(), out=self.Y1)
        if self._tails is not None:               # fused MLP tails: one launch per dependency level
            ta, tac = self._tails
            if not self._elu_on_load:
                self._elu(self.Y1[:, :nd])
            _chk(self.lib.go1ppo_tail_fwd(ctypes.byref(ta), _stream()), "go1ppo_tail_fwd")
            latent = self.Z["adaptation"][self.depth["adaptation"] - 1]
            if not self._elu_on_load:
                self._elu(self.Y1[:, nd:], latent, na)
            _chk(self.lib.go1ppo_tail_fwd(ctypes.byref(tac), _stream()), "go1ppo_tail_fwd")
            return self.Z["actor"][self.depth["actor"] - 1], self.Z["critic"][self.depth["critic"] - 1], latent
        self._elu(self.Y1[:, :nd])
        latent = self._tail("adaptation", self.Y1[:, :nd])
        self._elu(self.Y1[:, nd:], latent, na)
        with self._branch():
            value = self._tail("critic", self.Y1[:, nd + na:])
        mean = self._tail("actor", self.Y1[:, nd:nd + na])
        self._join()
        return mean, value, latent

    # ---- LDS-resident 256 -> 128 -> 64 ends ------------------------------------------------------------------------
    def _mlp2_fwd(self, key, items, elu_input=1):
        """items: [(net, x)] with x the 256-wide PRE-activation in front of the net's last two layers (activated in place);
        elu_input=0: x is already activated."""
        key = ("fwd", key, elu_input)
        if key not in self._mlp2_cache:
            arr = (Mlp2Fwd * len(items))()
            for a, (net, x) in zip(arr, items):
                d = self.depth[net]
                W2, W3, z2, out = self.P[f"{net}.{d - 2}.W"], self.P[f"{net}.{d - 1}.W"], self.Z[net][d - 2], self.Z[net][d - 1]
                a.x, a.W2, a.b2, a.W3, a.b3 = x.data_ptr(), W2.data_ptr(), self.P[f"{net}.{d - 2}.b"].data_ptr(), W3.data_ptr(), self.P[f"{net}.{d - 1}.b"].data_ptr()
                a.z2, a.out, a.rows, a.ld_x, a.ld_z2, a.ld_out, a.elu_input = z2.data_ptr(), out.data_ptr(), x.shape[0], _ld(x), _ld(z2), _ld(out), int(elu_input)
            self._mlp2_cache[key] = (arr, len(items))
        arr, n = self._mlp2_cache[key]
        _chk(self.lib.go1ppo_mlp2_fwd(arr, n, _stream()), "go1ppo_mlp2_fwd")

    def _mlp2_bwd(self, key, items):
        """items: [(net, h, d_x)]: h = the activated 256-wide input of the last two layers, d_x = where the gradient w.r.t.
        its pre-activation goes.  Reads dZ[net][last], writes dZ[net][last - 1] and d_x."""
        key = ("bwd", key)
        if key not in self._mlp2_cache:
            arr = (Mlp2Bwd * len(items))()
            for a, (net, h, d_x) in zip(arr, items):
                d = self.depth[net]
                d_out, z2, d_z2 = self.dZ[net][d - 1], self.Z[net][d - 2], self.dZ[net][d - 2]
                a.d_out, a.z2, a.h, a.W2, a.W3 = d_out.data_ptr(), z2.data_ptr(), h.data_ptr(), self.P[f"{net}.{d - 2}.W"].data_ptr(), self.P[f"{net}.{d - 1}.W"].data_ptr()
                a.d_z2, a.d_x, a.rows = d_z2.data_ptr(), d_x.data_ptr(), h.shape[0]
                a.ld_dout, a.ld_z2, a.ld_h, a.ld_dz2, a.ld_dx = _ld(d_out), _ld(z2), _ld(h), _ld(d_z2), _ld(d_x)
            self._mlp2_cache[key] = (arr, len(items))
        arr, n = self._mlp2_cache[key]
        _chk(self.lib.go1ppo_mlp2_bwd(arr, n, _stream()), "go1ppo_mlp2_bwd")

    def _forward_mlp2(self, x):
        P, Z, nd, na = self.P, self.Z, self.nd, self.na
        torch.mm(x, P["W1"].t(), out=self.Y1)
        self._mlp2_fwd("adaptation", [("adaptation", self.Y1[:, :nd])])
        latent = Z["adaptation"][2]
        self._elu(self.Y1[:, nd:], latent, na)
        if self._l1_nt and self._pair:
            # actor's and critic's 512 -> 256 layers as ONE launch: no graph fork / join around them
            gemm_nt_pair(self.lib, dict(a=self.Y1[:, nd:nd + na], b=P["actor.1.W"], c=Z["actor"][1], bias=P["actor.1.b"]),
                         dict(a=self.Y1[:, nd + na:], b=P["critic.1.W"], c=Z["critic"][1], bias=P["critic.1.b"]))
        elif self._l1_nt:
            # the 512 -> 256 layers on go1ppo_gemm_nt (bias in the epilogue; measured 16 us against hipBLASLt's 31 at 24576 rows)
            with self._branch():
                gemm_nt(self.lib, self.Y1[:, nd + na:], P["critic.1.W"], Z["critic"][1], P["critic.1.b"])
            gemm_nt(self.lib, self.Y1[:, nd:nd + na], P["actor.1.W"], Z["actor"][1], P["actor.1.b"])
        else:
            with self._branch():
                torch.addmm(P["critic.1.b"], self.Y1[:, nd + na:], P["critic.1.W"].t(), out=Z["critic"][1])
            torch.addmm(P["actor.1.b"], self.Y1[:, nd:nd + na], P["actor.1.W"].t(), out=Z["actor"][1])
        self._join()
        self._mlp2_fwd("ac", [("actor", Z["actor"][1]), ("critic", Z["critic"][1])])
        return Z["actor"][3], Z["critic"][3], latent

    def _backward_mlp2(self, x):
        nd, na = self.nd, self.na
        P, G, Z, dZ, Y1, dY1, dH1 = self.P, self.G, self.Z, self.dZ, self.Y1, self.dY1, self.dH1
        cols = {"adaptation": slice(0, nd), "actor": slice(nd, nd + na), "critic": slice(nd + na, self.n1)}
        self._mlp2_bwd("ac", [("actor", Z["actor"][1], dZ["actor"][1]), ("critic", Z["critic"][1], dZ["critic"][1])])
        for net in ("actor", "critic"):                     # the loss kernel already produced the heads' bias gradients
            self._wgrad(dZ[net][3], Z[net][2], G[f"{net}.3.W"], None)
            self._wgrad(dZ[net][2], Z[net][1], G[f"{net}.2.W"], G[f"{net}.2.b"])
            self._wgrad(dZ[net][1], Y1[:, cols[net]], G[f"{net}.1.W"], G[f"{net}.1.b"])
        if self._dgrad_nt and self._pair and self._wt_by_optimizer:
            gemm_nt_pair(self.lib, dict(a=dZ["actor"][1], b=self._WT["actor"], c=dY1[:, cols["actor"]], elu_bwd_of=Y1[:, cols["actor"]]),
                         dict(a=dZ["critic"][1], b=self._WT["critic"], c=dY1[:, cols["critic"]], elu_bwd_of=Y1[:, cols["critic"]]))
        elif self._dgrad_nt:
            # input gradient of the 512 -> 256 layers with the ELU-backward factor of the first layer folded into the GEMM's
            # epilogue (go1ppo_gemm_nt, epilogue 2): dY1 = (dz1 W) * elu'(h1) in one pass per net instead of a hipBLASLt GEMM
            # + an element-wise pass over (M x 512).  The kernel wants the weight K-contiguous: W^T, refreshed here (256 KB).
            with self._branch():
                if not self._wt_by_optimizer:
                    self._WT["critic"].copy_(P["critic.1.W"].t())
                gemm_nt(self.lib, dZ["critic"][1], self._WT["critic"], dY1[:, cols["critic"]], elu_bwd_of=Y1[:, cols["critic"]])
            if not self._wt_by_optimizer:
                self._WT["actor"].copy_(P["actor.1.W"].t())
            gemm_nt(self.lib, dZ["actor"][1], self._WT["actor"], dY1[:, cols["actor"]], elu_bwd_of=Y1[:, cols["actor"]])
        else:
            with self._branch():
                torch.mm(dZ["critic"][1], P["critic.1.W"], out=dH1["critic"])
                self._elu_bwd(dH1["critic"], Y1[:, cols["critic"]], None, out=dY1[:, cols["critic"]])
            torch.mm(dZ["actor"][1], P["actor.1.W"], out=dH1["actor"])
            self._elu_bwd(dH1["actor"], Y1[:, cols["actor"]], None, out=dY1[:, cols["actor"]])
        # actor first layer's latent columns: a1 += latent Wz^T
        latent, dlat = Z["adaptation"][2], dZ["adaptation"][2]
        dA1 = dY1[:, cols["actor"]]
        self._wgrad(dA1, latent, G["Wz"])
        torch.mm(dA1, P["Wz"], out=dlat)          # (13 us on hipBLASLt; a dedicated streaming kernel for the two useful columns measured 19)
        self._mlp2_bwd("adaptation", [("adaptation", Y1[:, :nd], dY1[:, :nd])])
        self._wgrad(dlat, Z["adaptation"][1], G["adaptation.2.W"], G["adaptation.2.b"])
        self._wgrad(dZ["adaptation"][1], Y1[:, :nd], G["adaptation.1.W"], G["adaptation.1.b"])
        self._join()
        self._planned_wgrads_beside(lambda: self._big_wgrad(dY1, x, G["W1"], self._w1_tmp, self._w1_tn[0]))

    def _forward_adaptation_mlp2(self, x):
        torch.mm(x, self.P["W1"][:self.nd].t(), out=self.Y1d)
        self._mlp2_fwd("adaptation_only", [("adaptation", self.Y1d)])
        return self.Z["adaptation"][2]

    def _backward_adaptation_mlp2(self, x):
        nd, d, G, Z, dZ = self.nd, self.dH1["adaptation"], self.G, self.Z, self.dZ
        self._mlp2_bwd("adaptation_only", [("adaptation", self.Y1d, d)])
        self._wgrad(dZ["adaptation"][2], Z["adaptation"][1], G["adaptation.2.W"], None)     # head bias: the MSE kernel's
        self._wgrad(dZ["adaptation"][1], self.Y1d, G["adaptation.1.W"], G["adaptation.1.b"])
        self._big_wgrad(d, x, G["W1"][:nd], self._w1_tmp[:, :nd], self._w1_tn[1])

    # ---- K-contiguous weight copies kept by the optimiser ------------------------------------------------------------
    def adam_transposes(self):
        """[(start element in the flat parameter, rows, cols, destination tensor)] of the transposed copies the backward pass reads;
        after FusedAdam.set_transposes(...) took them over the per-pass transpose-copy launches are dropped."""
        if not getattr(self, "_dgrad_nt", False):
            return []
        out = []
        for n in ("actor", "critic"):
            i = self.pol.index[f"{n}.1.W"]
            rows, cols = self.pol.blocks[i][1]
            out.append((self.pol.offsets[i], rows, cols, self._WT[n]))
        return out

    def refresh_transposes(self):
        """after the compute copy was rewritten from outside the optimiser (initial push, resume, sharded step's all-gather)"""
        if getattr(self, "_dgrad_nt", False):
            for n in ("actor", "critic"):
                self._WT[n].copy_(self.P[f"{n}.1.W"].t())

    # ---- two-stream helpers ------------------------------------------------------------------------------------
    def _branch(self):
        """context: work issued inside runs on the side stream, ordered after everything issued so far."""
        if self._side is None:
            import contextlib
            return contextlib.nullcontext()
        self._side.wait_stream(torch.cuda.current_stream())
        self._forked = True
        return torch.cuda.stream(self._side)

    def _join(self):
        if self._side is not None and getattr(self, "_forked", False):      # (nothing to wait for when no branch was opened since the last join)
            torch.cuda.current_stream().wait_stream(self._side)
            self._forked = False

    def forward_adaptation(self, x):
        if self._mlp2:
            return self._forward_adaptation_mlp2(x)
        torch.mm(x, self.P["W1"][:self.nd].t(), out=self.Y1d)
        self._elu(self.Y1d)
        if self._tails_ad is not None:
            _chk(self.lib.go1ppo_tail_fwd(ctypes.byref(self._tails_ad), _stream()), "go1ppo_tail_fwd")
            return self.Z["adaptation"][self.depth["adaptation"] - 1]
        return self._tail("adaptation", self.Y1d)

    # ---- backward ------------------------------------------------------------------------------------------
    def _tail_bwd(self, net, h0, dh0, head_bias_done=True):
        """Given dZ[net][last], accumulate the tail's parameter gradients and write d(loss)/d(h0) into dh0.
        Bias gradients (column sums of dZ) ride along with the weight-gradient kernel; the head's were already
        produced by the loss kernel unless head_bias_done is False."""
        P, G, d = self.P, self.G, self.depth[net]
        for li in range(d - 1, 0, -1):
            dz = self.dZ[net][li]
            h_in = self.Z[net][li - 1] if li > 1 else h0
            gb = None if (li == d - 1 and head_bias_done) else G[f"{net}.{li}.b"]
            self._wgrad(dz, h_in, G[f"{net}.{li}.W"], gb)
            out = self.dZ[net][li - 1] if li > 1 else dh0
            torch.mm(dz, P[f"{net}.{li}.W"], out=out)
            if li > 1:
                self._elu_bwd(out, h_in, None)

    def _big_wgrad(self, dY, x, gW, tmp, own_kernel):
        """first-layer weight gradient.  own_kernel: one more problem of the batched 128-tile launch (slabs, or atomics straight
        into the fp32 gradient); otherwise one (rows x M) @ (M x Kp) bf16 hipBLASLt GEMM (the fp32-output variants it
        offers for this shape are 3x slower) and one cast into the fp32 gradient."""
        zr, zc0, zc1 = self.priv_mask           # (gW starts at row 0 of W1: rows of the adaptation module, then the actor's)
        zr = min(zr, gW.shape[0])
        if own_kernel:
            self._wgrad(dY, x, gW, zero=(zr, zc0, zc1))
            return
        b = tmp.shape[0]
        # the partial products are summed into the fp32 gradient by go1ppo_sum_partials, which also writes the structural zeros
        # of the privileged-observation columns
        if b == 1 or not tmp.is_contiguous():
            torch.mm(dY.t(), x, out=tmp[0])
            part, count, stride = tmp[0], 1, gW.numel()
        else:
            torch.bmm(dY.view(b, dY.shape[0] // b, dY.shape[1]).transpose(1, 2), x.view(b, x.shape[0] // b, x.shape[1]), out=tmp)
            part, count, stride = tmp, b, tmp.stride(0)
        assert gW.is_contiguous() and part.stride(-1) == 1 and part.stride(-2) == gW.shape[1]
        if self._slabs and (self._recording is not None or self._batched):      # summed with the other slab pieces of the pass
            if self._rec_pieces is not None:
                self._rec_pieces.append(dict(begin=self._grad_offset(gW), count=gW.numel(), src=part, stride=stride, kind=2, slabs=count,
                                             cols=gW.shape[1], zero=(zr, zc0, zc1)))
            return
        _chk(self.lib.go1ppo_sum_partials(part.data_ptr(), count, stride, gW.shape[0], gW.shape[1], gW.data_ptr(), zr, zc0, zc1, _stream()),
             "go1ppo_sum_partials")

    def _grad_offset(self, view):
        off = (view.data_ptr() - self._grad.data_ptr()) // 4
        assert view.dtype == torch.float32 and view.is_contiguous() and 0 <= off and off + view.numel() <= self._grad.numel()
        return int(off)

    def last_pieces(self, kind):
        """(device table, count) of the slab pieces the last backward pass of this kind ("ppo" / "adaptation") left unsummed
        (defer_grad_sum), or None"""
        return self._last_pieces.get(kind)

    def _run_planned(self, key, fn):
        """first call: run `fn` recording its weight-gradient problems (nothing launched for them), build the device
        table; every call: run `fn` with the per-layer launches suppressed, then the one batched launch."""
        if key not in self._plans:
            self._recording, self._rec_pieces = [], []
            fn()
            rec, self._recording = self._recording, None
            pieces, self._rec_pieces = self._rec_pieces, None
            tab = (WgradProblem * len(rec))()
            for P, (dz, h, gW, gb, zero) in zip(tab, rec):
                P.dz, P.h, P.dW, P.bias_grad = dz.data_ptr(), h.data_ptr(), gW.data_ptr(), _ptr(gb)
                if zero is not None:
                    P.zero_n, P.zero_k0, P.zero_k1 = zero
                P.rows, P.ld_dz, P.ld_h, P.n, P.k, P.ldw = dz.shape[0], _ld(dz), _ld(h), dz.shape[1], h.shape[1], gW.shape[1]
            tn = self._wgrad_tn and all(P.rows % 64 == 0 and P.n % 8 == 0 and P.k % 8 == 0 for P in tab)
            plan = self.lib.go1ppo_wgrad_tn_plan if tn else self.lib.go1ppo_wgrad_plan
            if tn and self._slabs:       # the plan depends on WHETHER a problem has slabs (tile shape), not on where they are: a placeholder until they are allocated
                for P, (dz, h, gW, gb, zero) in zip(tab, rec):
                    P.partials, P.partial_stride = 16, gW.numel()
            total = plan(tab, len(rec))
            if total <= 0:
                raise RuntimeError(f"go1ppo_wgrad_plan failed with code {total}")
            if tn and self._slabs:
                # one fp32 slab per row chunk and problem (shared by the plans of the same pass: graph mode keeps one plan per
                # pre-gathered input block); zero-initialised and never written on the structural zeros
                for j, (P, (dz, h, gW, gb, zero)) in enumerate(zip(tab, rec)):
                    S = -(-P.rows // P.chunk_rows)
                    ws = self._slab_ws.get((key[0], j))
                    if ws is None or tuple(ws.shape) != (S, gW.numel()):
                        ws = self._slab_ws[(key[0], j)] = torch.zeros(S, gW.numel(), device=gW.device, dtype=torch.float32)
                    P.partials, P.partial_stride = ws.data_ptr(), gW.numel()
                    pieces.append(dict(begin=self._grad_offset(gW), count=gW.numel(), src=ws, stride=gW.numel(), kind=1, slabs=S, cols=gW.shape[1],
                                       zero=(0, 0, 0)))
                if plan(tab, len(rec)) != total:
                    raise RuntimeError("go1ppo_wgrad_tn_plan: the plan changed with the slab buffers attached")
            dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.Y1.device)
            ptab = None
            if pieces:
                pieces.sort(key=lambda q: q["begin"])
                arr = (GradPiece * len(pieces))()
                end = 0
                for A, q in zip(arr, pieces):
                    assert q["begin"] >= end and q["begin"] % 8 == 0 and q["count"] % 8 == 0 and q["stride"] % 8 == 0, "slab pieces: ascending, disjoint, multiples of 8"
                    end = q["begin"] + q["count"]
                    A.begin, A.count, A.src, A.stride, A.kind, A.slabs, A.cols = q["begin"], q["count"], q["src"].data_ptr(), q["stride"], q["kind"], q["slabs"], q["cols"]
                    A.zero_rows, A.zero_c0, A.zero_c1 = q["zero"]
                ptab = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.Y1.device), len(pieces), [q["src"] for q in pieces])
            self._plans[key] = (dev, len(rec), total, rec, tn, ptab)
            # the recording pass skipped the launches AND ran the rest of fn: its dgrad results are valid, only the
            # weight gradients are missing -> fall through to the batched launch
        else:
            self._batched, self._plan_key, self._plan_launched = True, key, False
            try:
                fn()
            finally:
                self._batched, self._plan_key = False, None
            if self._plan_launched:          # fn() put the batched launch on the side stream next to the first-layer GEMM
                self._finish_pieces(key)
                return
        self._launch_plan(key)
        self._finish_pieces(key)

    def _finish_pieces(self, key):
        ptab = self._plans[key][5]
        self._last_pieces[key[0]] = None
        if ptab is None:
            return
        if self.defer_grad_sum:
            self._last_pieces[key[0]] = ptab[:2]
        else:
            _chk(self.lib.go1ppo_grad_reduce(self._grad.data_ptr(), ptab[0].data_ptr(), ptab[1], _stream()), "go1ppo_grad_reduce")

    def _launch_plan(self, key):
        dev, count, total, _, tn, _ = self._plans[key]
        launch = self.lib.go1ppo_wgrad_tn_batched if tn else self.lib.go1ppo_wgrad_batched
        _chk(launch(dev.data_ptr(), count, total, _stream()), "go1ppo_wgrad_batched")

    def _planned_wgrads_beside(self, big):
        """GO1_WGRAD_OVERLAP=1: run `big` (the first-layer weight gradient on hipBLASLt) with the pass's batched small weight
        gradients on the side stream.  Measured (same box, alternating): 1.5 % SLOWER end to end — both fight for the L2 -> LDS
        path — so it is off; the sequential order stays the default."""
        key = getattr(self, "_plan_key", None)
        if key is None or self._side is None or os.environ.get("GO1_WGRAD_OVERLAP", "0") != "1":
            return big()
        with self._branch():
            self._launch_plan(key)
        self._plan_launched = True
        big()
        self._join()

    def backward(self, x):
        # the plan holds device pointers: one per input block (graph mode feeds a different pre-gathered block per mini-batch)
        self._run_planned(("ppo", x.data_ptr()), lambda: (self._backward_mlp2 if self._mlp2 else self._backward)(x))

    def backward_adaptation(self, x):
        self._run_planned(("adaptation", x.data_ptr()), lambda: (self._backward_adaptation_mlp2 if self._mlp2 else self._backward_adaptation)(x))

    def _backward(self, x):
        """After forward(x) and a loss kernel that filled dZ[actor][last], dZ[critic][last] (+ their bias / std
        gradients): everything else.  Bias gradients are ACCUMULATED into the (pre-zeroed) flat gradient; weight gradients too without
        slabs (`_slabs` off) — with slabs they are WRITTEN: by go1ppo_grad_reduce at the end of the pass, or by the optimiser's norm pass
        (`defer_grad_sum`), see `_run_planned`."""
        nd, na = self.nd, self.na
        G, Y1, dY1, dH1 = self.G, self.Y1, self.dY1, self.dH1
        cols = {"adaptation": slice(0, nd), "actor": slice(nd, nd + na), "critic": slice(nd + na, self.n1)}
        with self._branch():
            self._tail_bwd("critic", Y1[:, cols["critic"]], dH1["critic"])
            self._elu_bwd(dH1["critic"], Y1[:, cols["critic"]], None, out=dY1[:, cols["critic"]])
        self._tail_bwd("actor", Y1[:, cols["actor"]], dH1["actor"])
        self._elu_bwd(dH1["actor"], Y1[:, cols["actor"]], None, out=dY1[:, cols["actor"]])
        # actor first layer's latent columns: a1 += latent Wz^T
        last = self.depth["adaptation"] - 1
        latent, dlat = self.Z["adaptation"][last], self.dZ["adaptation"][last]
        dA1 = dY1[:, cols["actor"]]
        self._wgrad(dA1, latent, G["Wz"])
        torch.mm(dA1, self.P["Wz"], out=dlat)
        self._tail_bwd("adaptation", Y1[:, :nd], dH1["adaptation"], head_bias_done=False)
        self._elu_bwd(dH1["adaptation"], Y1[:, :nd], None, out=dY1[:, :nd])
        self._join()
        self._big_wgrad(dY1, x, G["W1"], self._w1_tmp, self._w1_tn[0])

    def _backward_adaptation(self, x):
        nd, d = self.nd, self.dH1["adaptation"]
        self._tail_bwd("adaptation", self.Y1d, d)
        self._elu_bwd(d, self.Y1d, None)
        self._big_wgrad(d, x, self.G["W1"][:nd], self._w1_tmp[:, :nd], self._w1_tn[1])

    # ---- losses ----------------------------------------------------------------------------------------------
    def ppo_loss(self, st, idx, std, g_std, A, kl, acc):
        """st: RolloutStorage; acc: fp32 [value_loss, surrogate, ...] accumulators; kl: fp32 scalar (pre-zeroed)."""
        la, lc = self.depth["actor"] - 1, self.depth["critic"] - 1
        flat = lambda t: t.flatten(0, 1)
        a = LossArgs()
        a.mean, a.value, a.std = self.Z["actor"][la].data_ptr(), self.Z["critic"][lc].data_ptr(), std.data_ptr()
        a.head_ld, a.num_actions, a.rows = HEAD, std.numel(), self.M
        a.idx = idx.data_ptr()
        for field, t in (("actions", st.actions), ("old_mu", st.mu), ("old_sigma", st.sigma), ("old_logp", st.actions_log_prob),
                         ("advantages", st.advantages), ("returns", st.returns), ("old_values", st.values)):
            t = flat(t)
            assert t.is_contiguous() and t.dtype == torch.float32
            setattr(a, field, t.data_ptr())
        a.clip_param, a.value_loss_coef, a.entropy_coef = A.clip_param, A.value_loss_coef, A.entropy_coef
        a.use_clipped_value_loss = int(A.use_clipped_value_loss)
        a.d_mean, a.d_value = self.dZ["actor"][la].data_ptr(), self.dZ["critic"][lc].data_ptr()
        a.d_std = g_std.data_ptr()
        a.d_mean_bias = self.G[f"actor.{la}.b"].data_ptr()
        a.d_value_bias = self.G[f"critic.{lc}.b"].data_ptr()
        a.kl = kl.data_ptr()
        a.value_loss = acc[0:].data_ptr()
        a.surrogate_loss = acc[1:].data_ptr()
        _chk(self.lib.go1ppo_loss(ctypes.byref(a), _stream()), "go1ppo_loss")

    def adaptation_loss(self, st, idx, num_train, selective, acc):
        last = self.depth["adaptation"] - 1
        pred, dpred = self.Z["adaptation"][last], self.dZ["adaptation"][last]
        target = st.privileged_observations.flatten(0, 1)
        assert target.is_contiguous() and target.dtype == torch.float32 and target.shape[1] == self.pol.npv
        _chk(self.lib.go1ppo_mse(pred.data_ptr(), HEAD, target.data_ptr(), self.pol.npv, idx.data_ptr(), self.M, num_train,
                                 int(selective), dpred.data_ptr(), self.G[f"adaptation.{last}.b"].data_ptr(),
                                 acc[2:].data_ptr(), acc[3:].data_ptr(), _stream()), "go1ppo_mse")


# ---- rollout glue ----------------------------------------------------------------------------------------------
def act(lib, mean, value, std, noise, st, s):
    """sample + log-prob + policy outputs into rollout-storage slot s (see go1ppo_act)."""
    _chk(lib.go1ppo_act(mean.data_ptr(), value.data_ptr(), _ld(mean), std.data_ptr(), std.numel(), mean.shape[0], noise.data_ptr(),
                        st.actions[s].data_ptr(), st.mu[s].data_ptr(), st.sigma[s].data_ptr(), st.values[s].data_ptr(),
                        st.actions_log_prob[s].data_ptr(), _stream()), "go1ppo_act")


def store_step(lib, st, s, rewards, dones, time_outs, env_bins, gamma):
    n = rewards.shape[0]
    for t, dt in ((rewards, torch.float32), (dones, torch.uint8)):
        assert t.is_contiguous() and t.dtype == dt and t.shape[0] == n
    if time_outs is not None:
        assert time_outs.is_contiguous() and time_outs.element_size() == 1
    if env_bins is not None:
        assert env_bins.is_contiguous() and env_bins.dtype == torch.int32
    _chk(lib.go1ppo_store_step(rewards.data_ptr(), dones.data_ptr(), _ptr(time_outs), _ptr(env_bins), st.values[s].data_ptr(), gamma, n,
                               st.rewards[s].data_ptr(), st.dones[s].data_ptr(), st.env_bins[s].data_ptr() if env_bins is not None else None,
                               _stream()), "go1ppo_store_step")


# ---- observation ring (RolloutStorage(ring=True)) ----------------------------------------------------------------
def ring_step(lib, st, s, obs, privileged_obs, obs_history, X):
    """rollout step s: (s == 0: the whole window from the environment's fp32 history into ring rows 0 .. H-1;) append the
    new observation, assemble the augmented inference rows X (N x Kp), store the fp32 obs / privileged obs of the step."""
    N, no, H = st.num_envs, st.obs_ring.shape[2], st.history_length
    npv = st.privileged_observations.shape[-1]
    for t in (obs, privileged_obs):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == N
    assert X.dtype == torch.bfloat16 and X.is_contiguous() and tuple(X.shape) == (N, st.padded_width)
    if s == 0:
        assert obs_history.dtype == torch.float32 and obs_history.stride(1) == 1 and tuple(obs_history.shape) == (N, H * no)
        _chk(lib.go1ppo_ring_snapshot(obs_history.data_ptr(), obs_history.stride(0), N, H, no, st.obs_ring.data_ptr(), _stream()),
             "go1ppo_ring_snapshot")
    dst = None if s == 0 else st.obs_ring[s + H - 1].data_ptr()
    _chk(lib.go1ppo_ring_step(obs.data_ptr(), privileged_obs.data_ptr(), dst, st.obs_ring[s].data_ptr(), N, H, no, npv, st.padded_width,
                              X.data_ptr(), st.observations[s].data_ptr(), st.privileged_observations[s].data_ptr(), _stream()),
         "go1ppo_ring_step")


def ring_gather(lib, st, idx, X):
    """X (len(idx) x Kp) <- augmented history rows of the storage entries idx (flat index s * N + n)."""
    N, no, H = st.num_envs, st.obs_ring.shape[2], st.history_length
    priv = st.privileged_observations
    assert idx.dtype == torch.int64 and idx.is_contiguous() and X.is_contiguous() and tuple(X.shape) == (idx.numel(), st.padded_width)
    assert priv.dtype == torch.float32 and priv.is_contiguous()
    _chk(lib.go1ppo_ring_gather(st.obs_ring.data_ptr(), priv.data_ptr(), idx.data_ptr(), idx.numel(), N, H, no, priv.shape[-1],
                                st.padded_width, X.data_ptr(), _stream()), "go1ppo_ring_gather")


def gae(lib, st, last_values, gamma, lam, stats):
    T, N = st.num_transitions_per_env, st.num_envs
    stats.zero_()
    _chk(lib.go1ppo_gae(st.rewards.data_ptr(), st.dones.data_ptr(), st.values.data_ptr(), last_values.data_ptr(), T, N, gamma, lam,
                        st.returns.data_ptr(), st.advantages.data_ptr(), stats.data_ptr(), _stream()), "go1ppo_gae")
    stats[2] = float(T * N)


def normalize(lib, st, stats):
    _chk(lib.go1ppo_normalize(st.advantages.data_ptr(), st.advantages.numel(), stats.data_ptr(), _stream()), "go1ppo_normalize")


# ---- optimiser step ------------------------------------------------------------------------------------------------
class FusedAdam:
    """torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay) over element ranges of the flat fp32 master
    parameter, with the gradient clip, the KL-adaptive learning rate and the refresh of the compute copies fused in
    (go1ppo_opt_prestep + go1ppo_opt_adam).  `ranges`: up to two (start, count) pairs."""

    def __init__(self, lib, master, body, std, n_body, lr, ranges=None, betas=(0.9, 0.999), eps=1e-8):
        self.lib, self.master, self.body, self.std, self.n_body = lib, master, body, std, n_body
        dev = master.device
        self.m, self.v = torch.zeros_like(master), torch.zeros_like(master)
        self.step = torch.zeros(1, device=dev)
        self.lr = torch.full((1,), float(lr), device=dev)
        self.partial = torch.zeros(lib.go1ppo_opt_partials(), device=dev)
        self.betas, self.eps = betas, eps
        r = list(ranges or [(0, master.numel())]) + [(0, 0)]
        self.r0, self.r1 = r[0], r[1]
        self.n_norm = max(a + b for a, b in r)          # elements the global norm runs over (padding slots excluded)
        self.extras = AdamExtras()
        self._keep = []

    def set_transposes(self, items):
        """items: [(start, rows, cols, dst bf16 (cols x rows) tensor)] — dst is rewritten whenever its weights are stepped"""
        e = self.extras
        assert len(items) <= 2
        e.num_transposes = len(items)
        self._keep = [t for *_, t in items]
        for T, (start, rows, cols, dst) in zip(e.transpose, items):
            assert dst.is_contiguous() and dst.dtype == torch.bfloat16 and tuple(dst.shape) == (cols, rows)
            T.start, T.rows, T.cols, T.dst = int(start), int(rows), int(cols), dst.data_ptr()

    def set_ranges(self, ranges):
        """restrict the step to these (start, count) element ranges (at most two) — e.g. one rank's slice of a sharded step"""
        r = list(ranges) + [(0, 0)]
        self.r0, self.r1 = r[0], r[1]

    def step_(self, gscale=1.0, max_norm=None, kl=None, kl_scale=1.0, desired_kl=0.01, lr_min=1e-5, lr_max=1e-2, zero_grad=False,
              zero_slot=None, between=None, pieces=None):
        """zero_grad: clear the visited gradient elements (and `zero_slot`, a one-element tensor) inside the Adam kernel.
        between: called between the two kernels with the per-block partial sums of the squared gradient norm — a sharded
        step all-reduces them there, so that every rank clips by the GLOBAL norm."""
        g = self.master.grad
        clip = max_norm is not None
        if pieces is not None:
            # pieces = FusedNet.last_pieces(kind): weight-gradient slabs the backward pass left unsummed — summed into g by the norm pass
            _chk(self.lib.go1ppo_opt_prestep_pieces(g.data_ptr(), self.n_norm, pieces[0].data_ptr(), pieces[1], gscale, self.partial.data_ptr(),
                                                    self.step.data_ptr(), self.lr.data_ptr(), _ptr(kl), kl_scale, desired_kl, lr_min, lr_max,
                                                    _stream()), "go1ppo_opt_prestep_pieces")
        else:
            _chk(self.lib.go1ppo_opt_prestep(g.data_ptr(), self.n_norm, gscale, self.partial.data_ptr() if clip else None,
                                             self.step.data_ptr(), self.lr.data_ptr(), _ptr(kl), kl_scale, desired_kl, lr_min, lr_max,
                                             _stream()), "go1ppo_opt_prestep")
        if between is not None:
            between(self.partial)
        _chk(self.lib.go1ppo_opt_adam(self.master.data_ptr(), g.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.r0[0], self.r0[1],
                                      self.r1[0], self.r1[1], gscale, self.partial.data_ptr() if clip else None,
                                      float(max_norm) if clip else 0.0, self.step.data_ptr(), self.lr.data_ptr(), self.betas[0],
                                      self.betas[1], self.eps, self.body.data_ptr(), self.n_body, self.std.data_ptr(), self.std.numel(), int(zero_grad),
                                      _ptr(zero_slot), ctypes.byref(self.extras), _stream()),
             "go1ppo_opt_adam")

"""Flat, functional training representation of the `ActorCritic` (MI355X-side plumbing, no reference analogue).

The nn.Module in actor_critic.py stays the public / checkpoint format (same state_dict keys as the reference).  For
training, its parameters are re-laid-out into ONE flat buffer so that

  * the first layers of the adaptation module, the actor and the critic — which all consume the 2100-wide
    observation history — are rows of a single (256+512+512) x Kp matrix W1, stored exactly as the GEMM wants it.
    Columns: [history (K) | 1 | privileged obs | 0-padding] to a multiple of 8, matching the rollout storage's
    augmented rows x' = [h, 1, p, 0]; so first-layer biases and the critic's privileged-input weights ride inside
    that one GEMM (W [h; z] + b = W_h h + W_z z + b·1);
  * output layers with 1 / 2 / 12 columns are stored zero-padded to 64 rows (GEMM libraries serve N=1..12 with very
    slow kernels); the padded rows never receive gradient;
  * autograd sees ONE leaf: parameters are `flat.split(...)` views, whose backward is a single concatenation — one
    gradient tensor, one norm for clipping, one (multi-)tensor Adam step, one RCCL all-reduce in data-parallel mode.

`pack()` / `unpack()` convert between the module and the flat layout (exact, no arithmetic).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

HEAD_COLS = 64


def _linears(seq):
    return [m for m in seq if isinstance(m, nn.Linear)]


class FlatPolicy:
    def __init__(self, ac, align=128):
        self.K = ac.num_obs_history
        self.npv = ac.num_privileged_obs
        # augmented row length, padded to the GEMM K-step: rows of X and W1 are then whole 128-B lines (hipBLASLt runs the
        # first-layer GEMMs 2x faster on 2112 columns than on 2104, and go1ppo_gemm_nt requires K % 64 == 0)
        self.Kp = -(-(self.K + 1 + self.npv) // 64) * 64
        self.act = type(ac.adaptation_module[1]) if len(ac.adaptation_module) > 1 else nn.ELU
        self.act_fn = {nn.ELU: F.elu, nn.ReLU: F.relu, nn.SELU: F.selu, nn.LeakyReLU: F.leaky_relu, nn.Tanh: torch.tanh,
                       nn.Sigmoid: torch.sigmoid}[self.act]
        self.nets = {"adaptation": _linears(ac.adaptation_module), "actor": _linears(ac.actor_body),
                     "critic": _linears(ac.critic_body)}
        self.first = [self.nets[n][0].out_features for n in ("adaptation", "actor", "critic")]
        self.out_dims = {n: self.nets[n][-1].out_features for n in self.nets}
        # ---- layout -------------------------------------------------------------------------------------
        self.blocks = []         # (name, shape)
        # order: the adaptation module's tail first, then W1 (whose first rows are the adaptation module's first
        # layer): everything the adaptation optimiser / its gradient all-reduce touches is ONE contiguous range
        def tail(n):
            lins = self.nets[n]
            for li, lin in enumerate(lins[1:], start=1):
                last = li == len(lins) - 1
                rows = max(lin.out_features, HEAD_COLS) if last else lin.out_features
                self.blocks.append((f"{n}.{li}.W", (rows, lin.in_features)))
                self.blocks.append((f"{n}.{li}.b", (rows,)))
        tail("adaptation")
        self.blocks.append(("W1", (sum(self.first), self.Kp)))
        assert self.npv <= HEAD_COLS
        self.blocks.append(("Wz", (self.first[1], HEAD_COLS)))      # columns >= npv stay exactly zero
        tail("actor")
        tail("critic")
        self.sizes, self.offsets, off = [], [], 0
        for _, shape in self.blocks:
            n = 1
            for d in shape:
                n *= d
            padded = -(-n // align) * align
            self.sizes.append(padded)
            self.offsets.append(off)
            off += padded
        self.numel = off
        self.index = {name: i for i, (name, _) in enumerate(self.blocks)}
        # [0, adaptation_numel): adaptation tail blocks + the adaptation rows of W1
        self.adaptation_numel = self.offsets[self.index["W1"]] + self.first[0] * self.Kp

    # ---- module <-> flat ---------------------------------------------------------------------------------
    def _block(self, flat, name):
        i = self.index[name]
        shape = self.blocks[i][1]
        n = 1
        for d in shape:
            n *= d
        return flat[self.offsets[i]:self.offsets[i] + n].view(shape)

    @torch.no_grad()
    def pack(self, ac, flat):
        """module parameters -> flat buffer (fp32)."""
        flat.zero_()
        K, npv = self.K, self.npv
        W1 = self._block(flat, "W1")
        r = 0
        for n in ("adaptation", "actor", "critic"):
            lin = self.nets[n][0]
            rows = lin.out_features
            W1[r:r + rows, :K].copy_(lin.weight[:, :K])
            W1[r:r + rows, K].copy_(lin.bias)
            if n == "critic":
                W1[r:r + rows, K + 1:K + 1 + npv].copy_(lin.weight[:, K:])
            elif n == "actor":
                self._block(flat, "Wz")[:, :npv].copy_(lin.weight[:, K:])
            r += rows
        for n, lins in self.nets.items():
            for li, lin in enumerate(lins[1:], start=1):
                self._block(flat, f"{n}.{li}.W")[:lin.out_features].copy_(lin.weight)
                self._block(flat, f"{n}.{li}.b")[:lin.out_features].copy_(lin.bias)

    @torch.no_grad()
    def unpack(self, flat, ac):
        """flat buffer -> module parameters (the export / checkpoint format)."""
        K, npv = self.K, self.npv
        W1 = self._block(flat, "W1")
        r = 0
        for n in ("adaptation", "actor", "critic"):
            lin = self.nets[n][0]
            rows = lin.out_features
            lin.weight[:, :K].copy_(W1[r:r + rows, :K])
            lin.bias.copy_(W1[r:r + rows, K])
            if n == "critic":
                lin.weight[:, K:].copy_(W1[r:r + rows, K + 1:K + 1 + npv])
            elif n == "actor":
                lin.weight[:, K:].copy_(self._block(flat, "Wz")[:, :npv])
            r += rows
        for n, lins in self.nets.items():
            for li, lin in enumerate(lins[1:], start=1):
                lin.weight.copy_(self._block(flat, f"{n}.{li}.W")[:lin.out_features])
                lin.bias.copy_(self._block(flat, f"{n}.{li}.b")[:lin.out_features])

    # ---- functional forward ---------------------------------------------------------------------------------
    def views(self, flat):
        """dict name -> parameter view; built with one split so that the backward is one concatenation."""
        parts = flat.split(self.sizes)
        out = {}
        for (name, shape), p in zip(self.blocks, parts):
            n = 1
            for d in shape:
                n *= d
            out[name] = (p if p.numel() == n else p.split([n, p.numel() - n])[0]).view(shape)
        return out

    def _tail(self, P, net, h):
        lins = self.nets[net]
        for li in range(1, len(lins)):
            h = F.linear(self.act_fn(h), P[f"{net}.{li}.W"], P[f"{net}.{li}.b"])
        n = self.out_dims[net]
        return h if h.shape[1] == n else h.split([n, h.shape[1] - n], dim=1)[0]

    def forward(self, flat, x, want_value=True, want_actor=True):
        """x: (M, Kp) augmented history rows [h, 1, privileged, 0] in flat's dtype.  Returns (mean, value, latent)."""
        P = self.views(flat)
        nd, na, nc = self.first
        if want_actor:
            W1 = P["W1"] if want_value else P["W1"].split([nd + na, nc])[0]
        else:
            W1 = P["W1"].split([nd, na + nc])[0]
        y = F.linear(x, W1)
        if not want_actor:
            return None, None, self._tail(P, "adaptation", y)
        ys = y.split([nd, na, nc] if want_value else [nd, na], dim=1)
        latent = self._tail(P, "adaptation", ys[0])
        a1 = ys[1]
        Wz = P["Wz"]
        for i in range(self.npv):
            a1 = torch.addcmul(a1, latent[:, i:i + 1], Wz[:, i])
        mean = self._tail(P, "actor", a1)
        value = self._tail(P, "critic", ys[2]) if want_value else None
        return mean, value, latent

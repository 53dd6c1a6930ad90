"""Observation-sharded logistic Gibbs over torch.distributed (RCCL on MI355X).

One process per GPU.  Rank k holds rows [idx0_k, idx0_k + N_k) of X.  Per sweep
(reference loop Code/C/Logit.hpp:426-450):

    shard.sweep_local(s)        psi, omega, partial PP_k = X_k' Omega_k X_k   (one pass over X_k)
    all_reduce(shard.pp())      the ONLY per-sweep exchange: P*P float64 over xGMI
    shard.draw_beta(s, c)       PP += P0, Cholesky, beta draw -- redundantly on every rank from
                                the same (seed, sweep) Philox stream, so beta needs no broadcast

X'kappa (constant over the chain, Logit.hpp:174-183) is all-reduced once at setup.
omega_i is keyed by the GLOBAL observation index, so the draws do not depend on
how many ranks share the rows.

`shard` is any object with the GibbsShard interface (bayeslogit_amd.device.GibbsShard
is the HIP one); the driver itself only sequences calls and collectives, which is
what the world_size-2 gloo tests exercise on CPU with a stand-in shard.
"""
import torch
import torch.distributed as dist


def shard_range(N, rank, world):
    """Contiguous rows of rank `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ReplicaDivergence(RuntimeError):
    """The ranks' redundantly drawn beta are no longer bit-identical."""


class DistGibbs:
    def __init__(self, shard, group=None, check_every=0, zero_copy=False):
        """check_every = k > 0: every k-th sweep verify that all ranks hold the same beta (one all-reduce of 2P
        doubles); 0: only when verify_replicas() is called (run() calls it once at the end).
        zero_copy: all-reduce the library's own P x P buffer in place (a torch view of foreign device memory) instead of a
        torch-owned staging tensor (two device copies of P*P doubles per sweep: microseconds).  Off by default: the collective
        libraries' stream bookkeeping is written for memory of torch's own allocator, and no multi-GPU node has run this yet."""
        self.shard = shard
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.check_every = int(check_every)
        self.zero_copy = bool(zero_copy)
        self._stage = {}

    def _all_reduce(self, t):
        if self.world <= 1:
            return
        if self.zero_copy:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return
        buf = self._stage.get(t.numel())
        if buf is None:
            buf = self._stage[t.numel()] = torch.empty(t.numel(), dtype=t.dtype, device=t.device)
        buf.copy_(t)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(buf)

    def setup(self, m0, P0, beta0=None):
        s = self.shard
        s.set_prior(m0, P0)
        if hasattr(s, "chain_start"):
            s.chain_start()
        s.set_bp_local()
        self._all_reduce(s.bp())
        s.finish_bp()
        if beta0 is not None:
            s.set_beta(beta0)

    def verify_replicas(self):
        """Every rank draws beta redundantly from the same (seed, sweep) stream and the same all-reduced PP, so the
        replicas must agree bit for bit; a collective that returned different bits on different ranks would let the
        chains drift apart silently.  max over ranks of (beta, -beta): equal halves <=> identical replicas."""
        if self.world == 1:
            return
        b = self.shard.beta()
        t = torch.cat([b, -b])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        P = b.numel()
        if not torch.equal(t[:P], -t[P:]):
            raise ReplicaDivergence("beta differs between ranks (the all-reduced X'Omega X was not bit-identical everywhere)")

    def sweep(self, sweep, constrain=1, w_out=None):
        s = self.shard
        s.sweep_local(sweep, w_out)
        self._all_reduce(s.pp())
        s.draw_beta(sweep, constrain)
        if self.check_every > 0 and (sweep + 1) % self.check_every == 0:
            self.verify_replicas()

    def run(self, samp, burn, constrain=1):
        """burn + samp sweeps; returns beta history (samp, P) as a CPU tensor."""
        s = self.shard
        hist = []
        sweep = 0
        for _ in range(burn):
            self.sweep(sweep, constrain)
            sweep += 1
        for _ in range(samp):
            self.sweep(sweep, constrain)
            hist.append(s.beta().clone())
            sweep += 1
        self.verify_replicas()
        return torch.stack(hist).cpu()

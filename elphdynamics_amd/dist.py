"""Multi-GPU plumbing for bench.py and multi-chain runs: one process per GPU, torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" for the CPU tests).

The hot path shards over INDEPENDENT CHAINS (one phonon configuration = one fermion matrix per rank): this
is how the reference itself parallelises (independent run-IDs, ElPhDynamics.jl:90-95) and what SURVEY.md
§8e recommends at these lattice sizes.  There is no data-path collective; the only communication is the
barrier around the timed region, the MAX-reduction of the elapsed time and the SUM of the work counters.
"""
import os

import numpy as np


class Comm:
    """Thin wrapper so that world_size == 1 needs no torch at all."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.torch = None
        self.dist = None
        self.device = None
        self.backend = None
        # ELPH_DIST_FORCE_INIT=1: a process group even at world size 1 (the collective transport of the sharded solve then runs its all-reduces
        # through RCCL on a one-GPU box: tests/test_gpu_shard.py)
        if self.world > 1 or os.environ.get("ELPH_DIST_FORCE_INIT") == "1":
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            # ELPH_DIST_BACKEND=gloo: rehearsal of the multi-rank driver on a box with fewer GPUs than ranks
            backend = backend or os.environ.get("ELPH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend="nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend=backend)

    def device_index(self):
        """GPU of this rank: LOCAL_RANK, unless ELPH_FORCE_DEVICE pins every rank to one device (rehearsals on a 1-GPU box)."""
        forced = os.environ.get("ELPH_FORCE_DEVICE")
        if forced is not None:
            return int(forced)
        return self.local_rank if self.world > 1 else 0

    # one independent chain per rank: distinct, reproducible seeds
    def chain_seed(self, base):
        return int(base) + 1009 * self.rank

    def barrier(self):
        if self.dist is not None:
            if self.backend == "nccl":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max(self, value):
        """MAX over ranks of a python float."""
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    # ---- small numpy-level collectives (halo slices, partial sums): a few KB each, staged through torch tensors
    def _to_tensor(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        return t.to(self.device) if self.backend == "nccl" else t

    def allgather(self, vec):
        """Concatenation over ranks of equally long float64 vectors (identical result on every rank)."""
        import numpy as np
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        if self.dist is None:
            return vec.copy()
        t = self._to_tensor(vec)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.concatenate([o.cpu().numpy() for o in out])

    def allgather_object(self, obj):
        """List over ranks of arbitrary (picklable) objects — ragged pieces of a result, not a hot-path call."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def ring_exchange(self, send_to_prev, send_to_next, recv_prev_n=None, recv_next_n=None):
        """Periodic ring: returns (received from previous rank, received from next rank).  The received lengths default
        to the symmetric case (what comes from the previous rank is as long as what goes to the next one)."""
        import numpy as np
        if self.dist is None:
            return np.array(send_to_next, copy=True), np.array(send_to_prev, copy=True)
        prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        sp, sn = self._to_tensor(send_to_prev), self._to_tensor(send_to_next)
        rp = self.torch.empty_like(sn) if recv_prev_n is None else self.torch.empty(int(recv_prev_n), dtype=sn.dtype, device=sn.device)
        rn = self.torch.empty_like(sp) if recv_next_n is None else self.torch.empty(int(recv_next_n), dtype=sp.dtype, device=sp.device)
        ops = [self.dist.P2POp(self.dist.isend, sp, prev), self.dist.P2POp(self.dist.isend, sn, nxt),
               self.dist.P2POp(self.dist.irecv, rp, prev), self.dist.P2POp(self.dist.irecv, rn, nxt)]
        if self.world == 2:      # prev == next: order the two messages by tag-free pairing (send order = receive order)
            ops = [self.dist.P2POp(self.dist.isend, sp, prev), self.dist.P2POp(self.dist.irecv, rn, nxt),
                   self.dist.P2POp(self.dist.isend, sn, nxt), self.dist.P2POp(self.dist.irecv, rp, prev)]
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        return rp.cpu().numpy(), rn.cpu().numpy()

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


class _HybridShared:
    """State the rank threads of one process share (HybridComm)."""

    def __init__(self, proc_comm, nthreads):
        import threading
        self.proc_comm, self.nthreads = proc_comm, int(nthreads)
        self.tb = threading.Barrier(self.nthreads)
        self.slots = [None] * self.nthreads
        self.result = None


class HybridComm:
    """Several ranks per process: rank = process rank * threads + thread index.  One thread per rank (every rank its own library
    handle, HIP stream and — on a multi-GPU host — device); between processes the process's `Comm` (gloo / RCCL), inside a
    process a thread barrier.  Only thread 0 ever touches torch.distributed.  What the in-library sharded solve needs from a
    host: all-gather of small objects (the 64-byte mailbox handles, the pieces of the result) and a barrier.
    Why it exists: a host language with one process and several GPUs (threads or tasks per device) drives `elph_shard_*` like
    this, and the one-GPU test box admits fewer processes on its card than the 8 ranks BASELINE.json names."""

    def __init__(self, shared, thread_index):
        self.sh, self.t = shared, int(thread_index)
        pc = shared.proc_comm
        self.world = pc.world * shared.nthreads
        self.rank = pc.rank * shared.nthreads + self.t
        self.local_rank = pc.local_rank * shared.nthreads + self.t
        self.backend = pc.backend

    @staticmethod
    def spawn(proc_comm, nthreads, target):
        """Run target(comm) on `nthreads` rank threads of this process; returns the list of results (exceptions re-raised)."""
        import threading
        shared = _HybridShared(proc_comm, nthreads)
        out, err = [None] * nthreads, [None] * nthreads

        def run(t):
            try:
                out[t] = target(HybridComm(shared, t))
            except BaseException as e:      # noqa: BLE001 — re-raised below; the other threads must not wait for this one forever
                err[t] = e
                shared.tb.abort()

        th = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for e in err:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out

    def device_index(self):
        forced = os.environ.get("ELPH_FORCE_DEVICE")
        return int(forced) if forced is not None else self.local_rank

    def barrier(self):
        self.sh.tb.wait()
        if self.t == 0:
            self.sh.proc_comm.barrier()
        self.sh.tb.wait()

    def allgather_object(self, obj):
        self.sh.slots[self.t] = obj
        self.sh.tb.wait()
        if self.t == 0:
            per_proc = self.sh.proc_comm.allgather_object(list(self.sh.slots))
            self.sh.result = [o for lst in per_proc for o in lst]
        self.sh.tb.wait()
        res = self.sh.result
        self.sh.tb.wait()                   # nobody overwrites slots / result before everybody has read them
        return res

    def max(self, value):
        return max(float(v) for v in self.allgather_object(float(value)))

    def sum(self, value):
        return float(sum(float(v) for v in self.allgather_object(float(value))))

    def close(self):
        self.barrier()


def timed_steps(comm, run_steps, steps):
    """The bench contract: barrier + synchronise, run exactly `steps` steps, synchronise + barrier, MAX over ranks.
    `run_steps(k)` must return only after the device has finished the k steps.  Returns (elapsed_max, total_work)
    where total_work sums the per-rank work counters run_steps returns (e.g. mat-vecs)."""
    import time
    comm.barrier()
    t0 = time.perf_counter()
    work = run_steps(steps)
    comm.barrier()
    elapsed = time.perf_counter() - t0
    return comm.max(elapsed), comm.sum(work)

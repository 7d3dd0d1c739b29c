"""Data-parallel DSAC-T update: one process + one engine per GPU, ONE collective per step.

The reference has no multi-worker learner; its only seam is `get_remote_update_info` /
`remote_update` (dsac_v2.py:107-138): gradient lists out, gradient lists in. This module slots one
all-reduce (RCCL over xGMI when the backend is "nccl") of the flat gradient arena
`[q1 | q2 | policy | log_alpha | mean_std1 | mean_std2]` between those two halves:

    every rank:  grads  = compute_grads(local minibatch of its own replay shard)     (local mean)
                 grads <- all_reduce(grads) / world                                  (one message)
                 apply_update()                                                      (identical on all ranks)

Local losses are local means, so the average over ranks is the gradient of the global-batch mean.
The two trailing floats carry the updated `mean_std` EMA so that the replicated state stays bitwise
identical across ranks without a second collective ("fast" mode).

`strict=True` adds the reference-exact variant: every rank's loss uses the GLOBAL
batch mean of the critics' std in the mean_std EMA, which costs a second, 2-float all-reduce between the
forward and the loss (engine.dp_forward / engine.std_sums / engine.dp_backward); the averaged gradients
then equal the single-process global-batch gradients to fp32 summation order, first step included.

`engine` is anything exposing `.grads` (flat torch tensor), `.dp_grads()`, `.dp_apply()` -- the
DsactEngine in production; the CPU tests drive the same coordinator with an oracle-backed stand-in
over the gloo backend.

Stream ordering: the engine's kernels run on the engine's own HIP stream, and a torch.distributed collective
orders itself against torch's *current* stream. Every collective (and the division that follows a SUM) is
therefore issued under `torch.cuda.stream(engine.torch_stream)` -- the engine's stream seen as a torch stream --
so "gradients complete -> all-reduce -> optimiser" is one stream-ordered chain without host synchronisation.
"""
import contextlib

import torch
import torch.distributed as dist


F_DATA_PARALLEL = 2   # dsact.h DSACT_F_DATA_PARALLEL


class DataParallelUpdater:
    def __init__(self, engine, group=None, broadcast_tensors=(), strict=False, overlap=False, native=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._avg = dist.get_backend(group) == "nccl"  # RCCL supports ReduceOp.AVG; gloo does not
        self.force_collective = False   # measurement aid: issue the all-reduce even with one rank
        self.strict = bool(strict)
        if self.strict:
            engine.dp_set_strict(True)
        # native: the library's own RCCL communicator (dsact_comm_init) instead of torch.distributed collectives: the
        # all-reduce is then a stream operation of the engine and the whole update can live in one hipGraph
        # (build_graph / run_graph). torch.distributed is still what ships the communicator id and the initial state.
        self.native = bool(native)
        if self.native:
            ids = [engine.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ids, src=0, group=group)
            engine.comm_init(self.rank, self.world, ids[0])
        # overlap: all-reduce the critics' segment (2/3 of the arena) while the actor's backward still runs
        self.overlap = bool(overlap) and not self.strict and hasattr(engine, "dp_grads_critic")
        self.graph_mode = False   # set by build_graph; run() without one takes the eager coordinator
        # replicas must start identical: rank 0's parameters / optimiser state win
        with self._on_engine_stream():
            for t in broadcast_tensors:
                dist.broadcast(t, src=0, group=group)

    def _on_engine_stream(self):
        ts = getattr(self.engine, "torch_stream", None)   # CPU stand-ins (gloo tests) have no stream
        return torch.cuda.stream(ts) if ts is not None else contextlib.nullcontext()

    def build_graph(self, steps_per_graph: int, fallback: bool = True) -> bool:
        """capture gather -> gradients -> RCCL all-reduce -> Adam/Polyak, `steps_per_graph` updates per hipGraph.

        The communicator is warmed up with ONE eager all-reduce first (RCCL sets its channels, proxies and buffers up on the
        first collective: that work does not belong inside a stream capture). With `fallback` a capture that fails on ANY
        rank (a runtime that cannot capture ncclAllReduce with real peers -- gpurun boxes have one GPU, so the first multi-GPU
        launch is also the first such capture) makes EVERY rank fall back to the eager coordinator (`step()`: the same
        library-owned communicator, one launch sequence per update) with a loud message instead of an exception on some
        ranks and a hang on the others. Returns True when the graph was captured; `graph_mode` tells which path `run()` takes."""
        if not self.native:
            raise RuntimeError("the graph-captured data-parallel update needs native=True (dsact_comm_init)")
        import sys

        ok, err = 1, None
        try:
            if self.world > 1 or self.force_collective:
                self.engine.dp_allreduce()          # warm-up on whatever the gradient arena holds (recomputed by every update)
                self.engine.sync()
            self.engine.graph_build(steps_per_graph, F_DATA_PARALLEL)
        except Exception as ex:   # DsactError (capture / instantiate / RCCL), or anything the runtime raises
            if not fallback:
                raise
            ok, err = 0, ex
        if self.world > 1:
            # (the flag lives where the GROUP's backend can reduce it: gloo groups are legitimate here -- the gradient traffic
            #  goes over the library's own RCCL communicator -- and cannot take a CUDA tensor)
            try:
                cpu_group = "gloo" in str(dist.get_backend(self.group)).lower()
            except Exception:
                cpu_group = True
            t = torch.tensor([ok], dtype=torch.int32, device="cpu" if cpu_group else getattr(self.engine, "device", "cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            all_ok = int(t.item())
        else:
            all_ok = ok
        self.graph_mode = bool(all_ok)
        if not all_ok:
            sys.stderr.write("dsact.dp: rank %d: the data-parallel hipGraph could NOT be captured (%s) -- falling back to the "
                             "EAGER coordinator on every rank (same RCCL communicator, one launch sequence per update)\n"
                             % (self.rank, repr(err) if err is not None else "another rank failed"))
            try:
                self.engine.sync()
            except Exception:
                pass
        return self.graph_mode

    def run(self, first_iteration: int, n_steps: int):
        """n_steps updates: graph replays when build_graph captured one, else the eager coordinator"""
        if getattr(self, "graph_mode", False):
            return self.run_graph(first_iteration, n_steps)
        self.engine.dp_begin(first_iteration)
        for _ in range(n_steps):
            self.step()

    def run_graph(self, first_iteration: int, n_steps: int):
        self.engine.graph_run(first_iteration, n_steps)

    def allreduce_grads(self):
        g = self.engine.grads
        if self.world == 1 and not self.force_collective:
            return
        if self.native:
            self.engine.dp_allreduce()
            return
        with self._on_engine_stream():
            if self._avg:
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                g.div_(self.world)

    def _reduce_async(self, t):
        if self._avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def step_overlapped(self):
        """critic half -> async all-reduce of grads[:n_c] on the collective's stream -> actor half ->
        all-reduce of the rest -> wait both -> apply. Same arithmetic as step()."""
        e = self.engine
        g, n_c = e.grads, e.critic_grad_count
        with self._on_engine_stream():
            e.dp_grads_critic()
            w1 = self._reduce_async(g[:n_c])      # starts behind the critic half on the engine's stream
            e.dp_grads_actor()
            w2 = self._reduce_async(g[n_c:])
            w1.wait()                             # the engine's stream waits for both results
            w2.wait()
            if not self._avg:
                g.div_(self.world)
        e.dp_apply()

    def step(self):
        if self.overlap:
            return self.step_overlapped()
        if self.strict:
            self.engine.dp_forward()
            if self.world > 1 or self.force_collective:
                with self._on_engine_stream():
                    dist.all_reduce(self.engine.std_sums, op=dist.ReduceOp.SUM, group=self.group)
            self.engine.dp_backward()
        else:
            self.engine.dp_grads()
        self.allreduce_grads()
        self.engine.dp_apply()

"""Row-sharded search across ranks: one process per GPU, torch.distributed (nccl == RCCL on ROCm).

The corpus never crosses xGMI: rank g owns rows [bounds[g], bounds[g+1]); every rank scores the
same query batch against its shard (fp64-exact per-shard top-k), then ONE all-gather moves
Q*k*(4+8)+4Q bytes per rank and every rank merges the gathered lists on its own device
(yams_scan_merge_topk_device).  The collective and the merge of batch i run on a side stream
(`GatherPipeline`); with several lanes per rank the caller keeps the device's sweep gate closed behind
a batch's sweep until its collective is enqueued (Accel.set_sweep_hold / release_sweep_hold): a filter
sweep is a persistent grid that owns every CU, and an RCCL kernel waiting for a CU under it would keep
its peers on the other GPUs spinning ("the fence").  torch is plumbing here: tensors,
streams and the collective."""
from __future__ import annotations

import os
import time


class CollectiveTimeout(RuntimeError):
    """An exchange (all-gather + merge) did not complete within GatherPipeline.watchdog_s."""


class Deadline:
    """`with Deadline("what", seconds):` — when the block has not finished in time, ONE line of diagnosis goes to stderr
    and the process exits with status 3 (torchrun then takes the other ranks down).  For the phases of an unattended
    multi-GPU run that can only hang, never fail: rendezvous, the communicator's first collective, a barrier, a run of
    steps whose exchange a peer never joins.  `on_expire(message)` replaces the exit (tests)."""

    def __init__(self, what: str, seconds: float, on_expire=None, detail=None):
        self.what, self.seconds, self.on_expire, self.detail = what, float(seconds), on_expire, detail
        self._t = None

    def _fire(self):
        import sys
        extra = ""
        if self.detail is not None:
            try:
                extra = " | " + str(self.detail())
            except Exception as e:          # noqa: BLE001
                extra = f" | (detail failed: {e!r})"
        msg = (f"[yams_amd watchdog] rank {os.environ.get('RANK', '0')}/{os.environ.get('WORLD_SIZE', '1')}: '{self.what}' "
               f"not finished after {self.seconds:.0f} s{extra} — aborting instead of waiting for the caller's timeout")
        if self.on_expire is not None:
            self.on_expire(msg)
            return
        sys.stderr.write(msg + "\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        import threading
        self._t = threading.Timer(self.seconds, self._fire)
        self._t.daemon = True
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()
        return False


def shard_bounds(n_rows: int, world: int) -> list[int]:
    """Contiguous row ranges: shard g owns rows [n*g/world, n*(g+1)/world)."""
    return [n_rows * g // world for g in range(world + 1)]


def init_from_env(backend: str | None = None, force: bool = False):
    """Rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.  force: form the process
    group for a single rank too (the one-GPU RCCL test: communicator, collective and merge of a world of one)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def launch_ranks(script: str, nproc: int, argv: list[str], capture: bool = False, timeout: float | None = None):
    """Start `nproc` ranks of `script` on this node under torch.distributed.run (one rank per GPU;
    rendezvous on 127.0.0.1 and a free port).  bench.py uses it when `--gpus N` is given without a
    torchrun environment; the CPU tests use it for the gloo jobs.  Returns the CompletedProcess."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    return subprocess.run(cmd, env=env, capture_output=capture, text=capture, timeout=timeout)


def _layout(spec):
    """Byte layout of one rank's record: the (name, nbytes) parts back to back, 16-byte aligned."""
    off, lay = 0, []
    for name, nb in spec:
        lay.append((name, off, nb))
        off += (nb + 15) & ~15
    return lay, off


class GatherPipeline:
    """The step after the per-shard scan, `depth` batches in flight.

    Slot s = batch i % depth owns one packed record per rank — [scores f32 [Q,k] | rows i64 [Q,k] |
    counts i32 [Q] (| dist f32 [Q,k])] — that the scan writes IN PLACE (`local(s)` hands out the
    views), so nothing is copied before the collective.  `launch(s)` issues, on a side stream,
    ONE all-gather of the record (RCCL: all_gather_into_tensor; gloo dry runs: the same call on a
    host copy) and then `merge_fn(gathered views, merged views)` — the k-way merge kernel on a
    context bound to that side stream.  The caller's next sweep runs meanwhile on its own stream;
    `wait(s)` blocks the host until slot s is merged (before its record is reused, or to read it).
    With world == 1 there is no collective and no merge: the merged views ARE the local views — unless
    `force_collective` asks for them anyway (a process group of one rank: what a one-GPU box can run of RCCL)."""

    def __init__(self, nq: int, k: int, device, with_dist: bool = False, depth: int = 2, force_collective: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.backend = dist.get_backend() if dist.is_initialized() else None
        self.active = dist.is_initialized() and (self.world > 1 or force_collective)   # collective + merge after every scan
        self.nq, self.k, self.depth, self.device = nq, k, depth, device
        spec = [("scores", nq * k * 4), ("rows", nq * k * 8), ("counts", nq * 4)]
        if with_dist:
            spec.append(("dist", nq * k * 4))
        self.lay, self.rec_bytes = _layout(spec)
        self.dtypes = {"scores": torch.float32, "rows": torch.int64, "counts": torch.int32, "dist": torch.float32}
        self.shapes = {"scores": (nq, k), "rows": (nq, k), "counts": (nq,), "dist": (nq, k)}
        self.rec = [torch.zeros(self.rec_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        on_gpu = device.type == "cuda"
        self.side = torch.cuda.Stream(device=device) if on_gpu and self.active else None
        if self.active:
            self.gathered = [torch.zeros((self.world, self.rec_bytes), dtype=torch.uint8, device=device) for _ in range(depth)]
            self.merged = [torch.zeros(self.rec_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
            self.done = [torch.cuda.Event(enable_timing=True) if on_gpu else None for _ in range(depth)]
            self.begun = [torch.cuda.Event(enable_timing=True) if on_gpu else None for _ in range(depth)]
            self.busy = [False] * depth
            self._keep = [None] * depth
        self.merge_fn = None
        # what the N > 1 bench line reports about the exchange: device time from the head of the all-gather to the
        # tail of the merge on the side stream (events), per batch; and a deadline after which a collective that
        # has not completed is declared stuck (CollectiveTimeout) instead of eating the caller's own timeout
        self.exchange_ms_sum, self.exchange_ms_max, self.exchanges = 0.0, 0.0, 0
        self.launched = 0
        self.watchdog_s = 30.0

    # -- views ---------------------------------------------------------------------------------------
    def _views(self, buf, lead=()):
        out = {}
        for name, off, nb in self.lay:
            t = buf[..., off:off + nb]
            if lead:                # [world][record] -> dense [world][...] per field (the merge ABI's layout)
                t = t.contiguous()
            out[name] = t.view(self.dtypes[name]).view(tuple(lead) + self.shapes[name])
        return out

    def local(self, slot):
        """Tensors the scan of this slot's batch writes (views into the slot's record)."""
        return self._views(self.rec[slot])

    def result(self, slot):
        """Merged top-k of the slot's batch (valid after wait(slot))."""
        return self.local(slot) if not self.active else self._views(self.merged[slot])

    def side_stream_ptr(self):
        return self.side.cuda_stream if self.side is not None else None

    # -- the step -------------------------------------------------------------------------------------
    def launch(self, slot):
        """Collective + merge of the slot's batch.  The scan that filled the record has completed on
        the host's view (yams_scan_topk_device synchronises its stream before returning)."""
        if not self.active:
            return
        torch, dist = self.torch, self.dist
        rec, out = self.rec[slot], self.gathered[slot]
        if self.backend == "gloo":      # CPU tests / single-GPU dry runs: the collective runs on the host
            t_h = time.perf_counter()
            src = rec.cpu()
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src)
            out.copy_(torch.stack(parts, 0))
            g = self._views(out, (self.world,))
            if self.side is not None:
                torch.cuda.current_stream().synchronize()
            self.merge_fn(g, self._views(self.merged[slot]))
            self._keep[slot] = g        # (allocated on the current stream, read on the side stream)
            if self.done[slot] is not None:
                self.done[slot].record(self.side)
            self._host_ms = (time.perf_counter() - t_h) * 1e3
            self.busy[slot] = True
            self.launched += 1
            return
        with torch.cuda.stream(self.side):
            self.begun[slot].record(self.side)
            work = dist.all_gather_into_tensor(out, rec, async_op=True)
            work.wait()                 # the side stream waits for RCCL's stream; the host does not
            self.merge_fn(self._views(out, (self.world,)), self._views(self.merged[slot]))
            self.done[slot].record(self.side)
        self.busy[slot] = True
        self.launched += 1

    def wait(self, slot):
        if not self.active or not self.busy[slot]:
            return
        ms = None
        if self.done[slot] is not None:
            ev = self.done[slot]
            if not ev.query():          # poll under a deadline: a collective whose peer never arrives must not hang the job
                # (sleeping, not spinning: the other lane's host thread needs the interpreter to enqueue its batch; a slot is
                #  normally complete long before it is waited for — the loop runs at the drain of a run of steps only)
                deadline = time.monotonic() + self.watchdog_s
                while not ev.query():
                    if time.monotonic() > deadline:
                        raise CollectiveTimeout(
                            f"all-gather + merge of slot {slot} not complete after {self.watchdog_s:.0f} s "
                            f"(backend {self.backend}, world {self.world}, rank {self.dist.get_rank()}, {self.rec_bytes} B per rank, "
                            f"{self.exchanges} exchanges completed before it): a peer rank is missing or the link is down")
                    time.sleep(0.0001)
            if self.backend != "gloo":
                ms = self.begun[slot].elapsed_time(ev)
        if ms is None:
            ms = getattr(self, "_host_ms", None)
        if ms is not None:
            self.exchange_ms_sum += ms; self.exchange_ms_max = max(self.exchange_ms_max, ms); self.exchanges += 1
        self.busy[slot] = False

    def reset_exchange_stats(self):
        self.exchange_ms_sum, self.exchange_ms_max, self.exchanges, self.launched = 0.0, 0.0, 0, 0

    def exchange_stats(self):
        """{exchanges, exchange_ms (mean), exchange_ms_max}: device time of all-gather + merge per batch on the side stream
        (gloo dry runs: host time of the staged collective + merge launch)."""
        n = self.exchanges
        return {"exchanges": n, "collectives": self.launched, "exchange_ms": self.exchange_ms_sum / n if n else None,
                "exchange_ms_max": self.exchange_ms_max if n else None}

    def drain(self):
        for s in range(self.depth):
            self.wait(s)
        if self.side is not None:
            self.side.synchronize()

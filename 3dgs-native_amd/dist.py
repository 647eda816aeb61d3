"""
View-parallel training support: one process per GPU, every rank holds the
full Gaussian set and renders its own camera view; the only exchange is ONE all-reduce per step over
the flat float32 gradient arena that backward() returns (`grads["_arena"]`, 59 floats per Gaussian:
mean3D | scale | rot | opacity | shs -- the five arrays the reference hands its optimizer,
train.py:1047-1051).  The reference itself is single-view, single-device (train.py:928).

Factored exchange (`exchange_factored` + `sh_gradients_from_views`): 48 of those 59 floats are the SH gradient, which for
one view is the outer product of the 16 SH basis values of the view direction with a 3-float colour gradient
(backward.py:95-255).  So instead of all-reducing 236 B per Gaussian, ranks all-reduce the other 11 floats (44 B) and
all-GATHER each view's 3 floats (+ its camera position); every rank then rebuilds the summed SH gradient locally with the
same per-view products.  Per GPU that moves 2(V-1)/V * 44 B + (V-1) * 12 B per Gaussian over xGMI instead of
2(V-1)/V * 236 B -- 161 MB instead of 413 MB at V = 8 and 1 M Gaussians -- and the backward no longer writes 192 B of SH
gradient per Gaussian.  The result equals the all-reduced gradient up to the order of the float sum over views.

backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

ARENA_FLOATS = 59
SMALL_ARENA_FLOATS = 11      # mean3D | scale | rot | opacity (backward(..., sh_gradient="factored"))
MAX_VIEWS_PER_CALL = 16      # GSR_MAX_VIEWS of include/gsr.h


def init_from_env(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # dmabuf IPC is what this pool's host driver supports.  The ROCr runtime reads HSA_ENABLE_IPC_MODE_LEGACY ONCE, when it
    # initialises -- i.e. at the process's first HIP call -- so the place to set it is the top of the entry script, before torch is
    # imported (bench.py, examples/train.py, tests/dist_worker.py and launch.py all do).  Setting it here only helps a rank that
    # has not touched the GPU yet; one that has, and did not inherit the variable, is told so instead of failing later inside RCCL.
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        if torch.cuda.is_initialized():
            import warnings
            warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY was not set before the HIP runtime started: RCCL's peer-memory sharing may fail "
                          "with hipIpcGetMemHandle: invalid argument.  Export HSA_ENABLE_IPC_MODE_LEGACY=0 before the first HIP call.")
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return dist.get_rank(), dist.get_world_size()


class ViewStreams:
    """Several views per GPU in flight at once (config #4 on fewer GPUs than views).  One view's
    launch- and latency-bound sort chain leaves most of the chip idle; issued on separate HIP streams, another view's blend
    kernels fill it (bench.py --views-per-step: 790 -> 935 Mpixels/s at three views in round 1).  The library keeps its scratch
    per stream, so the calls are independent; results equal the serial ones (float-atomic order aside).

        vs = ViewStreams(3)
        outs = vs.map(render_and_backward, views)      # outs[i] = fn(views[i]); all work is ordered into the caller's stream

    Item i runs on stream i % k.  Every stream first waits for the caller's stream (so inputs written there -- the parameters an
    optimizer step just updated -- are seen, and blocks the caching allocator recycles from a previous round are not reused
    early), and the caller's stream waits for all of them before `map` returns: tensors made inside `fn` may be used on the
    caller's stream right away."""

    def __init__(self, k, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(k)))]

    def map(self, fn, items):
        items = list(items)
        if len(self.streams) == 1 or len(items) <= 1:
            return [fn(it) for it in items]           # nothing to overlap: stay on the caller's stream
        main = torch.cuda.current_stream(self.device)
        used = self.streams[:min(len(self.streams), len(items))]
        for st in used:
            st.wait_stream(main)
        outs = []
        for i, it in enumerate(items):
            with torch.cuda.stream(self.streams[i % len(self.streams)]):
                outs.append(fn(it))
        for st in used:
            main.wait_stream(st)
        return outs


def views_for_rank(num_views, rank, world_size):
    """Indices of the camera views rank `rank` renders: round-robin, so any world size divides the batch
    as evenly as possible and no view is rendered twice."""
    return list(range(rank, num_views, world_size))


def arena_offsets(n, small=False):
    """Float offsets of the segments mean3D | scale | rot | opacity [| shs] inside one flat arena, each segment starting on a
    multiple of 4 floats: the kernels write dL_drot and the SH rows as 16-byte vectors, and include/gsr.h asks for 16-byte
    aligned array pointers.  Returns the starts plus the total length ([5] small, [6] dense); for N % 4 == 0 these are the
    plain 3N | 3N | 4N | N | 48N offsets.  The <= 3 padding floats between segments ride along in the collective."""
    sizes = [3 * n, 3 * n, 4 * n, n] + ([] if small else [48 * n])
    offs, o = [], 0
    for sz in sizes:
        offs.append(o)
        o = (o + sz + 3) & ~3
    return offs + [offs[-1] + sizes[-1]]


def arena_size(n, small=False):
    return arena_offsets(n, small)[-1]


def _n_of_arena(arena, small):
    per = SMALL_ARENA_FLOATS if small else ARENA_FLOATS
    n = arena.numel() // per          # padding is < 16 floats in total, so it can only push the quotient up for tiny n
    while n > 0 and arena_size(n, small) > arena.numel():
        n -= 1
    return n


def arena_views(arena, n):
    """Split a flat dense arena (arena_size(n) floats) into the five gradient arrays (views, no copies)."""
    o = arena_offsets(n)
    return {"dL_dmean3D": arena[o[0]:o[0] + 3 * n].view(n, 3), "dL_dscale": arena[o[1]:o[1] + 3 * n].view(n, 3),
            "dL_drot": arena[o[2]:o[2] + 4 * n].view(n, 4), "dL_dopacity": arena[o[3]:o[3] + n],
            "dL_dshs": arena[o[4]:o[4] + 48 * n].view(n * 16, 3)}


def reduce_gradients(arena, world_size=None, average=True):
    """In-place sum (or mean) of the gradient arena over all ranks: a single collective."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return arena
    ws = dist.get_world_size() if world_size is None else world_size
    if average and arena.is_cuda and dist.get_backend() == "nccl":
        # RCCL scales inside the collective (ncclAvg): no second 236 B/Gaussian pass over the arena
        dist.all_reduce(arena, op=dist.ReduceOp.AVG)
        return arena
    dist.all_reduce(arena, op=dist.ReduceOp.SUM)
    if average:
        arena.mul_(1.0 / ws)   # mean keeps densify_grad_threshold semantics (reference config.py:54)
    return arena


def small_arena_views(arena, n):
    """Split the 11-float arena of the factored mode (arena_size(n, small=True) floats; views, no copies)."""
    o = arena_offsets(n, small=True)
    return {"dL_dmean3D": arena[o[0]:o[0] + 3 * n].view(n, 3), "dL_dscale": arena[o[1]:o[1] + 3 * n].view(n, 3),
            "dL_drot": arena[o[2]:o[2] + 4 * n].view(n, 4), "dL_dopacity": arena[o[3]:o[3] + n]}


def exchange_factored(arena, payload, average=True):
    """The two collectives of the factored exchange.  `arena` ([11*N], from backward(..., sh_gradient="factored")) is summed
    (or averaged) in place over ranks; `payload` ([3*N + 4]) is all-gathered.  Returns the [V, 3*N + 4] tensor of all
    views' payloads, rank order (V = world size; a single process gets its own payload back as V = 1)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return payload.unsqueeze(0)
    world = dist.get_world_size()
    gathered = torch.empty((world, payload.numel()), dtype=payload.dtype, device=payload.device)
    work = dist.all_gather_into_tensor(gathered.view(-1), payload.contiguous(), async_op=True)   # the two run back to back
    reduce_gradients(arena, world, average)
    work.wait()
    return gathered


class FactoredExchange:
    """The factored exchange with its collectives overlapped with compute (one view per rank per step):

        ex = FactoredExchange()
        g = backward(..., sh_gradient="factored", on_payload=ex.start_gather)   # all-gather runs beside the per-Gaussian half
        grads = ex.finish(g, means3D, degree)                                   # all-reduce runs beside the SH rebuild

    RCCL executes the two collectives in issue order on its own stream: the all-gather of the payloads (known after the blend
    half of the backward) overlaps `geom_backward_kernel`; the all-reduce of the 11-float arena (known after it) overlaps the
    rebuild kernel, which needs only the gathered payloads.  Single process: no collectives, same results."""

    def __init__(self, timing=False):
        self._gathered = None
        self._work = None
        # timing=True: five events per step on the compute stream (payload ready | before the gather wait | after it | rebuild
        # launched, before the reduce wait | after it); `timings_ms()` turns them into the stalls the exchange really costs
        self.timing = bool(timing)
        self._events = []
        self._cur = None

    def _mark(self):
        if self.timing and self._cur is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._cur.append(e)

    def start_gather(self, payload):
        self._cur = [] if (self.timing and payload.is_cuda) else None
        self._mark()
        if not dist.is_initialized() or dist.get_world_size() == 1:
            self._gathered, self._work = payload.unsqueeze(0), None
            return
        world = dist.get_world_size()
        self._gathered = torch.empty((world, payload.numel()), dtype=payload.dtype, device=payload.device)
        self._work = dist.all_gather_into_tensor(self._gathered.view(-1), payload.contiguous(), async_op=True)

    def timings_ms(self):
        """Per recorded step (after a device synchronize): the compute stream's time from `payload ready` to the end of the
        exchange (`total`: the per-Gaussian backward half runs inside it), its stall at the all-gather wait, the rebuild kernel
        + the all-reduce wait (`rebuild_and_reduce_wait`), and the all-reduce stall that remained after the rebuild."""
        out = []
        for ev in self._events:
            if len(ev) == 5:
                out.append({"total": ev[0].elapsed_time(ev[4]), "gather_wait": ev[1].elapsed_time(ev[2]),
                            "rebuild_and_reduce_wait": ev[2].elapsed_time(ev[4]), "reduce_wait": ev[3].elapsed_time(ev[4])})
        return out

    @staticmethod
    def bytes_per_rank(n, world):
        """Bytes one rank sends (= receives) per step: the 11-float arena through a bandwidth-optimal all-reduce
        (2 (V-1)/V of it) and its 3-float payload + camera position to each of the V-1 peers."""
        reduce_b = 2.0 * (world - 1) / world * 4 * arena_size(n, small=True)
        gather_b = (world - 1) * 4 * (3 * n + 4)
        return {"all_reduce": int(reduce_b), "all_gather": int(gather_b), "total": int(reduce_b + gather_b),
                "dense_all_reduce_instead": int(2.0 * (world - 1) / world * 4 * arena_size(n))}

    def finish(self, grads, means3D, degree=3, average=True, out=None):
        """Returns the dict of the five averaged (or summed) optimizer gradients."""
        arena = grads["_arena"]
        n = _n_of_arena(arena, small=True)
        reduce_work = None
        if dist.is_initialized() and dist.get_world_size() > 1:
            if average and arena.is_cuda and dist.get_backend() == "nccl":
                reduce_work = dist.all_reduce(arena, op=dist.ReduceOp.AVG, async_op=True)
            else:
                reduce_work = dist.all_reduce(arena, op=dist.ReduceOp.SUM, async_op=True)
        self._mark()
        if self._work is not None:
            self._work.wait()
        self._mark()
        res = {"dL_dshs": sh_gradients_from_views(means3D, self._gathered, degree, average=average, out=out)}
        self._mark()
        if reduce_work is not None:
            reduce_work.wait()
            if average and not (arena.is_cuda and dist.get_backend() == "nccl"):
                arena.mul_(1.0 / dist.get_world_size())
        self._mark()
        if self._cur is not None:
            self._events.append(self._cur)
            self._cur = None
        res.update(small_arena_views(arena, n))
        self._gathered = self._work = None
        return res


def sh_gradients_from_views(means3D, payloads, degree=3, average=True, out=None, scale=None):
    """dL_dshs [N*16, 3] summed (or averaged) over the views whose payloads are the rows of `payloads` ([V, 3*N + 4]) --
    the HIP kernel behind gsr_sh_grad_from_views.  `payloads` may also be a list of V separate [3*N + 4] tensors.
    `scale` overrides the factor (1/V when averaging) -- e.g. 1/batch when some rows are zero padding."""
    from . import _host, _lib
    rows = [payloads[v] for v in range(len(payloads))]
    if not rows:
        raise ValueError("sh_gradients_from_views: no view payloads")
    n = int(means3D.shape[0])
    dev = means3D.device
    for r in rows:
        if not (r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and r.numel() == 3 * n + 4):
            raise ValueError(f"sh_gradients_from_views: each payload must be a contiguous float32 device tensor of {3 * n + 4} elements")
    if not (means3D.is_cuda and means3D.dtype == torch.float32 and means3D.is_contiguous()):
        raise ValueError("sh_gradients_from_views: means3D must be a contiguous float32 device tensor")
    if out is None:
        out = torch.empty((n * 16, 3), dtype=torch.float32, device=dev)
    factor = float(scale) if scale is not None else (1.0 / len(rows) if average else 1.0)
    with _host.on_device(dev):
        # the kernel takes at most GSR_MAX_VIEWS payload rows per call: larger batches are rebuilt in chunks and summed
        tmp = None
        for c0 in range(0, len(rows), MAX_VIEWS_PER_CALL):
            chunk = rows[c0:c0 + MAX_VIEWS_PER_CALL]
            if c0 == 0:
                dst = out
            else:
                dst = tmp = torch.empty_like(out) if tmp is None else tmp
            ptrs = (C.c_void_p * len(chunk))(*[r.data_ptr() for r in chunk])
            _lib.check(_lib.lib().gsr_sh_grad_from_views(n, _host.ptr(means3D), int(degree), len(chunk), ptrs, factor,
                                                         _host.ptr(dst), _host.stream_ptr(dev)))
            if c0:
                out.add_(dst)
    return out

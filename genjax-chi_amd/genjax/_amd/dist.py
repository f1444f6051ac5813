"""Multi-device execution: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on the GPU box; "gloo" in the CPU tests).  The reference has no distributed layer at all;
this is a new design around two properties of the single-device path:

  * keys are counter-based and indexed by GLOBAL particle slot, and
  * weight sums are exact integers,

so a population sharded over G ranks produces bit-identical particles, ancestors and log-Z for
every G — the exchange is pure data movement, never a change of arithmetic.

ImportanceK: rank r runs a row-aligned block of slots with no data-path communication.  The importance
kernel emits row-anchored partial sums of its own log-weights; one tiny kernel folds them into a
65-word record (anchor exponent + 64 exact integer buckets) and ONE all-gather of that
record per pass yields the global log-normaliser — no second pass over the log-weights, no all-reduce
pair, and the merged result is bit-identical for every number of ranks.

Bootstrap SMC: resampling is global.  Per step each rank (1) resamples + propagates + weights its own
output slots reading the GLOBAL previous population — only the source tiles that feed its slots —
(2) all-reduce(max) of the per-tile maxima, (3) computes its tiles' fixed-point masses and all-gathers
them (8 B per 1024 particles), (4) the ancestor shuffle: ancestors are monotone, so each rank needs ONE
contiguous global source range, which every rank derives from the tile masses; it arrives in place by
grouped send/recv from the ranks that own it (an all-to-all-v with no packing and no count exchange).
`exchange="allgather"` keeps the simpler all-gather of the whole population.
"""

from __future__ import annotations

import torch

from . import abi, prng, workloads as W
from .ops import Ops


def _dist():
    import torch.distributed as dist

    return dist


ROW = 256  # particles per row of the row-anchored partial sums (gjx_num_max_partials)


def shard_rows(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """(first slot, slot count) of rank's block when the population is split at row boundaries (what
    keeps the row-anchored log-normaliser independent of the number of ranks)."""
    rows = -(-n_total // ROW)
    r0, r1 = rows * rank // world, rows * (rank + 1) // world
    first = r0 * ROW
    return first, min(r1 * ROW, n_total) - first


class BatchedImportance:
    """Sharded ImportanceK passes with *bucketed, overlapped* collectives.

    A pass = the importance kernel, which leaves its row sums in slot b; one `gjx_lse_rows_batch` launch
    per batch folds them into the shard's 65-word records, a [batch, 65] block.  After `batch` passes the block is all-gathered ONCE, asynchronously (RCCL runs on
    its own stream), while the next batch already computes into the other block of a double buffer; the
    wait is deferred until that block is reused or its results are read.  A small-message RCCL collective
    costs tens of microseconds — more than the 28 us kernel — so this is the xGMI analogue of gradient
    bucketing with compute/communication overlap: per pass the exchange costs 1/batch of one collective
    and none of the device's time.  Passes reuse one set of trace buffers (48 MB at 1e6 particles)."""

    def __init__(self, ops: Ops, wl, batch: int = 8, world: int | None = None, depth: int = 2,
                 always_exchange: bool = False, passes: int = 1):
        dist = _dist()
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        if self.world > 1 and (wl.first % ROW or (wl.n % ROW and wl.first + wl.n != wl.n_total)):
            raise ValueError("shards must be split at 256-particle row boundaries (dist.shard_rows)")
        self.ops, self.wl, self.batch, self.depth = ops, wl, batch, depth
        self.passes = passes  # independent passes per importance launch (gjx_importance_run_batch)
        self.prep = wl.prepare(fold_batch=batch, passes=passes)
        dev, words = ops.device(), abi.LSE_RECORD_WORDS
        self.local = [torch.zeros((batch, words), dtype=torch.int64, device=dev) for _ in range(depth)]
        self.exchange = self.world > 1 or always_exchange  # a one-rank group still goes through the collective
        self.gathered = [torch.zeros((self.world, batch, words), dtype=torch.int64, device=dev) if self.exchange
                         else t.view(1, batch, words) for t in self.local]
        import ctypes as C

        self._rec = [C.c_void_p(t.data_ptr()) for t in self.local]
        self._work = [None] * depth
        self._count = [0] * depth
        self._slot = 0

    def run(self, count: int | None = None, on_launch=None) -> int:
        """Enqueue `count` (<= batch) passes and the exchange of their records; returns the buffer
        index to hand to `results`.  `on_launch(phase, n_passes)` brackets every importance launch."""
        count = self.batch if count is None else count
        d, self._slot = self._slot, (self._slot + 1) % self.depth
        self.wait(d)  # the block's previous exchange must have read it before it is overwritten
        st, done = self.ops.stream(), 0
        while done < count:
            c = min(self.passes, count - done)
            if on_launch:
                on_launch(0, c)
            if self.passes == 1:
                self.prep.launch_importance(st, done)  # row sums into slot `done`
            else:
                self.prep.launch_passes(done, c, st)  # c passes, row sums into slots done .. done + c - 1
            if on_launch:
                on_launch(1, c)
            done += c
        self.prep.launch_fold(count, st, self._rec[d])  # one launch folds the batch into its records
        if self.exchange:
            self._work[d] = _dist().all_gather_into_tensor(self.gathered[d].view(-1, self.gathered[d].shape[-1]),
                                                            self.local[d], async_op=True)
        self._count[d] = count
        return d

    def wait(self, d: int | None = None):
        for i in (range(self.depth) if d is None else (d,)):
            if self._work[i] is not None:
                self._work[i].wait()
                self._work[i] = None

    def results(self, d: int):
        """(lse f32[count], e i32[count], q i64[count]) of buffer d's passes (device tensors)."""
        self.wait(d)
        return self.ops.lse_combine(self.gathered[d], self._count[d])

    def log_z(self, d: int, b: int = 0) -> float:
        _, e, q = self.results(d)
        return self.ops.log_z_from_rows(e[b], q[b], self.wl.n_total)


def importance_log_z(ops: Ops, wl: "W.Gaussian10"):
    """One sharded ImportanceK pass -> (log_z float64, local logw, e, q).  `wl` was built with the
    rank's (first, n_local) from `shard_rows` and the global n_total."""
    pipe = BatchedImportance(ops, wl, batch=1, depth=1)
    d = pipe.run(1)
    _, e, q = pipe.results(d)
    return ops.log_z_from_rows(e[0], q[0], wl.n_total), pipe.prep.logw, e, q


def merged_tile_masses(recs):
    """Tile records (int64 [tiles, 2]: word 0 = S_t, low half of word 1 = e_t; gjx_tile_rec) -> (e, masses uint64
    [tiles]): the merge every consumer of the records performs: e = max e_t, M_t = S_t >> (e - e_t)."""
    import numpy as np

    from . import abi

    r = np.ascontiguousarray(np.asarray(recs.cpu() if hasattr(recs, "cpu") else recs)).view(np.uint64)
    r = r.reshape(r.shape[0], -1)
    s = r[:, 0].copy()
    et = (r[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32).astype(np.int64)
    live = et != abi.TILE_EMPTY
    e = int(et[live].max()) if live.any() else abi.TILE_EMPTY
    d = np.where(live, e - et, 64)
    m = np.where(d < 64, s >> np.minimum(d, 63).astype(np.uint64), np.uint64(0)).astype(np.uint64)
    return e, m


def needed_tile_ranges(tile_sums, n_total: int, tile: int, world: int):
    """For every rank of a filter sharded into equal contiguous blocks: the half-open range of SOURCE tiles that
    can own one of the rank's output slots at the next systematic resampling — int64 [world, 2].  Host (numpy)
    statement of `gjx_smc_source_ranges`, which the sharded filter runs on the device; kept as its cross-check.

    Source tile b owns the comb teeth between `teeth_below(prefix_b)` and `teeth_below(prefix_{b+1})`
    (gjx_device.hpp), `teeth_below(C) = ceil(C * (N / Q) - u0)` in float64 with the comb offset u0 in [0, 1).
    The same float64 products are formed here from the exact tile masses; u0 is bounded instead of derived, so
    the range is the exact one or one tile wider at either end.  Both bounds are monotone in b, hence the tiles a
    rank needs are contiguous."""
    import numpy as np

    q = np.ascontiguousarray(np.asarray(tile_sums)).view(np.uint64)
    nt = q.size
    prefix = np.zeros(nt + 1, dtype=np.uint64)
    np.cumsum(q, out=prefix[1:])
    if prefix[-1] == 0:  # no mass at all: the population is kept, every block's sources are its own tiles
        per = (n_total // world) // tile
        return np.array([[j * per, (j + 1) * per] for j in range(world)], dtype=np.int64)
    scale = np.float64(n_total) / np.float64(prefix[-1])
    teeth = np.minimum(np.ceil(prefix.astype(np.float64) * scale), np.float64(n_total))  # u0 = 0: the upper bound
    lo_b = np.maximum(teeth[:-1] - 1.0, 0.0)  # u0 -> 1
    hi_b = teeth[1:].copy()
    hi_b[-1] = n_total  # the last particle closes the comb
    n_local = n_total // world
    out = np.empty((world, 2), dtype=np.int64)
    for j in range(world):
        out[j, 0] = np.searchsorted(hi_b, np.float64(j * n_local), side="right")
        out[j, 1] = np.searchsorted(lo_b, np.float64((j + 1) * n_local), side="left")
    return out


class TorchComm:
    """The exchanges of a sharded filter over `torch.distributed` (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self, rank: int, world: int, always: bool = False):
        """`always`: issue the collectives even in a one-rank group (rehearsal of the N > 1 calls on a one-GPU box)."""
        self.rank, self.world, self.always = rank, world, always

    def all_gather(self, full: torch.Tensor, lo: int, hi: int):
        if self.world == 1 and not self.always:
            return
        local = full[lo:hi]
        if full.device.type != "cuda":
            local = local.clone()  # gloo: keep input and output distinct
        _dist().all_gather_into_tensor(full, local)

    def exchange(self, cols, sends, recvs, units=None):
        """Grouped point-to-point transfers of global slices of `cols`: `sends` / `recvs` = [(peer, a, b)] in PARTICLES; column
        c holds one row per units[c] particles (1: per-particle columns; the tile size: the per-tile sub-prefixes).  A slice
        keeps its global position on both sides, so nothing is packed or unpacked."""
        dist = _dist()
        units = units or [1] * len(cols)
        p2p = [dist.P2POp(dist.isend, c[a // u:b // u], peer) for peer, a, b in sends for c, u in zip(cols, units)]
        p2p += [dist.P2POp(dist.irecv, c[a // u:b // u], peer) for peer, a, b in recvs for c, u in zip(cols, units)]
        if p2p:
            for w in dist.batch_isend_irecv(p2p):
                w.wait()  # stream-ordered on RCCL; blocking on gloo


class ThreadComm:
    """`world` virtual ranks as threads of ONE process sharing one device and one stream: the same exchange
    protocol with host barriers in place of collectives.  The enqueue order on the shared stream is the
    execution order, so a barrier between "everyone enqueued its writes" and "everyone enqueues its reads" is
    all the synchronisation the copies need.  Used to run the sharded protocol on the HIP kernels of a one-GPU
    box (tests); not a transport."""

    class Shared:
        def __init__(self, world: int):
            import threading

            self.world, self.barrier, self.slots = world, threading.Barrier(world), [None] * world

    def __init__(self, shared: "ThreadComm.Shared", rank: int):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _post(self, obj):
        self.sh.slots[self.rank] = obj
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        return got

    def all_gather(self, full: torch.Tensor, lo: int, hi: int):
        got = self._post((full, lo, hi))
        for r, (src, a, b) in enumerate(got):
            if r != self.rank:
                full[a:b].copy_(src[a:b])
        self.sh.barrier.wait()

    def exchange(self, cols, sends, recvs, units=None):
        units = units or [1] * len(cols)
        got = self._post(cols)
        for peer, a, b in recvs:
            for mine, theirs, u in zip(cols, got[peer], units):
                mine[a // u:b // u].copy_(theirs[a // u:b // u])
        self.sh.barrier.wait()


class NativeComm:
    """`gjx_comm` (include/gjx.h): the library's own communicator.  `NativeComm.rccl(ops, rank, world)` builds an RCCL
    communicator (the 128-byte id travels through the torch.distributed store / broadcast); `NativeComm.local_group(ops,
    world)` gives `world` virtual ranks for threads of this process (tests)."""

    def __init__(self, ops: Ops, handle, group=None):
        self.ops, self.handle, self._group = ops, handle, group
        self.rank = int(ops.lib.call("gjx_comm_rank", handle))
        self.world = int(ops.lib.call("gjx_comm_world", handle))

    @staticmethod
    def rccl(ops: Ops, rank: int, world: int) -> "NativeComm":
        import ctypes as C

        from . import abi

        buf = torch.zeros(abi.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            ops.lib.call("gjx_comm_unique_id", C.c_void_p(buf.data_ptr()))
        if world > 1:
            dist = _dist()
            dev_buf = buf.to(ops.device()) if dist.get_backend() == "nccl" else buf
            dist.broadcast(dev_buf, 0)
            buf = dev_buf.cpu()
        h = C.c_void_p()
        ops.lib.call("gjx_comm_init_rccl", C.c_void_p(buf.data_ptr()), rank, world, C.byref(h))
        return NativeComm(ops, h)

    @staticmethod
    def over_torch(ops: Ops, rank: int, world: int, always: bool = False) -> "NativeComm":
        """`gjx_comm_init_callbacks`: the native driver with `torch.distributed` as its transport — the all-gather and the
        grouped send/recv of the process group the program already has (gloo on CPU: how `gjx_smc_sharded_run_*` is
        tested across REAL processes; nccl = RCCL on GPUs).  The callbacks wrap the raw buffers as byte tensors."""
        import ctypes as C

        from . import abi

        cuda = ops.device().type == "cuda"

        def view(ptr: int, nbytes: int) -> torch.Tensor:
            if cuda:
                class _Dev:  # a device buffer as an array the tensor constructor understands
                    __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

                return torch.as_tensor(_Dev(), device=ops.device())
            return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

        def allgather(_user, full, nbytes, _s):
            try:
                dist = _dist()
                t = view(full, world * nbytes)
                local = t[rank * nbytes:(rank + 1) * nbytes]
                dist.all_gather_into_tensor(t, local if cuda else local.clone())
                return 0
            except Exception as ex:  # an exception must not unwind through the C driver
                print(f"[gjx] all-gather callback failed: {ex!r}", flush=True)
                return -4

        def exchange(_user, cols, elems, units, n_cols, sends, ns, recvs, nr, _s):
            try:
                dist = _dist()
                ops_ = []
                for segs, n, fn in ((sends, ns, dist.isend), (recvs, nr, dist.irecv)):
                    for i in range(n):
                        sg = segs[i]
                        for c in range(n_cols):
                            el, un = int(elems[c]), int(units[c])
                            a, b = int(sg.a) // un * el, int(sg.b) // un * el
                            ops_.append(dist.P2POp(fn, view(int(cols[c]) + a, b - a), int(sg.peer)))
                if ops_:
                    for w in dist.batch_isend_irecv(ops_):
                        w.wait()
                return 0
            except Exception as ex:
                print(f"[gjx] exchange callback failed: {ex!r}", flush=True)
                return -4

        def sync(_user, _s):
            if cuda:
                torch.cuda.current_stream().synchronize()
            return 0

        cbs = (abi.ALLGATHER_FN(allgather), abi.EXCHANGE_FN(exchange), abi.STREAM_SYNC_FN(sync))
        h = C.c_void_p()
        ops.lib.call("gjx_comm_init_callbacks", rank, world, cbs[0], cbs[1], cbs[2], None, C.byref(h))
        nc = NativeComm(ops, h)
        nc._callbacks = cbs  # the C side keeps the function pointers: keep the Python objects alive as long
        return nc

    @staticmethod
    def local_group(ops: Ops, world: int) -> list:
        import ctypes as C

        g = C.c_void_p()
        ops.lib.call("gjx_comm_group_create", world, C.byref(g))
        owner = _GroupOwner(ops, g)
        out = []
        for r in range(world):
            h = C.c_void_p()
            ops.lib.call("gjx_comm_init_local", g, r, C.byref(h))
            out.append(NativeComm(ops, h, owner))
        return out

    @staticmethod
    def peers(ops: Ops, arena: "PeerArena", group=None, wait_launch: bool = False, timeout_ms: int = 0) -> "NativeComm":
        """`gjx_comm_init_peers`: the peer transport over `arena`.  `group`: the handle owner of a `gjx_comm_group` when the
        ranks are virtual ranks (threads) of this process sharing one stream; `wait_launch`: ranks that share a device as
        separate processes."""
        import ctypes as C

        st = arena.peers_struct(timeout_ms)
        h = C.c_void_p()
        ops.lib.call("gjx_comm_init_peers", C.byref(st), group.handle if group is not None else None, 1 if wait_launch else 0,
                     C.byref(h))
        nc = NativeComm(ops, h, group)
        nc._arena = arena
        return nc

    @staticmethod
    def peers_virtual(ops: Ops, arenas: list, timeout_ms: int = 0) -> list:
        """One peer-transport communicator per virtual rank of this process (threads sharing one stream)."""
        import ctypes as C

        g = C.c_void_p()
        ops.lib.call("gjx_comm_group_create", len(arenas), C.byref(g))
        owner = _GroupOwner(ops, g)
        return [NativeComm.peers(ops, a, owner, False, timeout_ms) for a in arenas]

    def lse_combine(self, records: torch.Tensor):
        """records int64[n_batch, 65] of this rank's shard -> (lse f32[n_batch], e i32[n_batch], q i64[n_batch]) of the
        whole population, the same on every rank (`gjx_comm_lse_combine`)."""
        from . import abi

        nb = records.shape[0]
        gathered = self.ops.empty((self.world, nb, abi.LSE_RECORD_WORDS), torch.int64)
        lse, e, q = self.ops.empty(nb, torch.float32), self.ops.empty(nb, torch.int32), self.ops.empty(nb, torch.int64)
        self.ops.lib.call("gjx_comm_lse_combine", self.handle, self.ops._p(records), nb, self.ops._p(gathered), self.ops._p(e),
                          self.ops._p(q), self.ops._p(lse), self.ops.stream())
        self._keep = gathered
        return lse, e, q

    def __del__(self):
        try:
            if self.handle:
                self.ops.lib.call("gjx_comm_destroy", self.handle)
                self.handle = None
        except Exception:
            pass


class PeerArena:
    """One rank's ARENA of the peer transport (gjx.h gjx_smc_peers): both populations of a sharded filter, the rank's arrival
    words and its error word, carved from ONE block of device memory at offsets that depend only on the filter's shape — so
    the same array sits at the same offset in every rank's arena and `address on rank o = address here + delta[o]`.

    `block`: a uint8 tensor of `PeerArena.nbytes(...)` bytes that the peers can address (a slice of one allocation for virtual
    ranks of one process; an allocation shared through hipIpc for ranks that are processes).  The arrival words must be zero
    when the arena is first used and are never reset afterwards (they only grow)."""

    HEAD = 256  # flags: int64[8] at offset 0; error: int32 at offset 128

    def __init__(self, block: torch.Tensor, delta: list[int], rank: int, world: int):
        self.block, self.delta, self.rank, self.world = block, list(delta), rank, world
        self.flags = block[0:64].view(torch.int64)
        self.error = block[128:132].view(torch.int32)
        self._pops = None

    @staticmethod
    def nbytes(ops: Ops, n_total: int, state_dtypes: list, adaptive: bool) -> int:
        from .ops import BlockCarver, SmcPopulation

        cv = BlockCarver(None)
        cv.off = PeerArena.HEAD
        for _ in range(2):
            SmcPopulation(ops, n_total, state_dtypes, adaptive, True, cv)
        return (cv.off + 4095) // 4096 * 4096

    def pops(self, ops: Ops, n_total: int, state_dtypes: list, adaptive: bool):
        from .ops import BlockCarver, SmcPopulation

        cv = BlockCarver(self.block)
        cv.off = PeerArena.HEAD
        self._pops = [SmcPopulation(ops, n_total, state_dtypes, adaptive, True, cv) for _ in range(2)]
        return self._pops

    def peers_struct(self, timeout_ms: int = 0):
        from . import abi

        p = abi.SmcPeers()
        p.world, p.rank = self.world, self.rank
        for o in range(self.world):
            p.delta[o] = int(self.delta[o])
        p.flags, p.error = self.flags.data_ptr(), self.error.data_ptr()
        p.timeout_ms = int(timeout_ms)
        return p

    def check(self, ops: Ops):
        if ops.device().type == "cuda":
            torch.cuda.current_stream().synchronize()
        if int(self.error.cpu()[0]) != 0:
            raise RuntimeError("peer transport: a step's wait for its peers timed out (gjx_smc_peers.error)")

    @staticmethod
    def ipc(ops: Ops, rank: int, world: int, n_total: int, state_dtypes: list, adaptive: bool, fine_grained: bool = False) -> "PeerArena":
        """The arena of one rank per PROCESS (one process per GPU, or processes sharing a device): allocated with hipMalloc,
        zeroed, handed to the peers through hipIpcGetMemHandle — the 64-byte handles travel over the `torch.distributed`
        process group the program already has — and every peer's arena mapped with hipIpcOpenMemHandle (peer access over xGMI
        when it lives on another device).  Collective over the group: every rank calls it with the same shape."""
        # (the arrival words are polled with system-scope loads behind system-scope acquires and written with system-scope
        # release stores: correct by the memory model for any device memory a peer can map)
        import ctypes as C

        dist = _dist()

        class IpcHandle(C.Structure):  # hipIpcMemHandle_t: 64 opaque bytes, passed by value
            _fields_ = [("reserved", C.c_char * 64)]

        hip = C.CDLL("libamdhip64.so")
        hip.hipIpcGetMemHandle.argtypes = [C.POINTER(IpcHandle), C.c_void_p]
        hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), IpcHandle, C.c_uint]
        nb = PeerArena.nbytes(ops, n_total, state_dtypes, adaptive)
        # every rank reaches every collective below whatever fails locally: failures are VOTED on, so that no rank is left
        # waiting in a collective its peer never enters
        mine, err, h = C.c_void_p(), None, IpcHandle()
        # fine_grained: hipExtMallocWithFlags(hipDeviceMallocFinegrained) — device memory that stays coherent with peers' accesses
        # without cache maintenance (slower for the bulk columns; the fallback when a peer's arrival words written into ordinary
        # device memory do not become visible to a polling kernel on some platform)
        if fine_grained:
            hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
            rc_alloc = hip.hipExtMallocWithFlags(C.byref(mine), C.c_size_t(nb), 0x1)
        else:
            rc_alloc = hip.hipMalloc(C.byref(mine), C.c_size_t(nb))
        if rc_alloc != 0 or hip.hipMemset(mine, 0, C.c_size_t(nb)) != 0 or hip.hipDeviceSynchronize() != 0:
            err = "hipMalloc / hipMemset failed"
        elif hip.hipIpcGetMemHandle(C.byref(h), mine) != 0:
            err = "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this driver)"
        handles = [None] * world
        dist.all_gather_object(handles, None if err else bytes(h))
        ptrs = []
        if err is None and any(x is None for x in handles):
            err = "a peer could not export its arena"
        for o in range(world):
            if err is not None:
                break
            if o == rank:
                ptrs.append(mine.value)
                continue
            p_ = C.c_void_p()
            if hip.hipIpcOpenMemHandle(C.byref(p_), IpcHandle.from_buffer_copy(handles[o]), 1) != 0:  # hipIpcMemLazyEnablePeerAccess
                err = f"hipIpcOpenMemHandle failed for rank {o}"
            ptrs.append(p_.value)
        votes = [None] * world
        dist.all_gather_object(votes, err)  # (also the barrier: every arena is zeroed and mapped before anybody signals into it)
        bad = [f"rank {r}: {v}" for r, v in enumerate(votes) if v is not None]
        if bad:
            raise RuntimeError("peer arena: " + "; ".join(bad))

        class _Dev:
            __cuda_array_interface__ = {"shape": (nb,), "typestr": "|u1", "data": (mine.value, False), "version": 2}

        block = torch.as_tensor(_Dev(), device=ops.device())
        arena = PeerArena(block, [ptrs[o] - mine.value for o in range(world)], rank, world)
        arena._ipc = (hip, mine, ptrs)  # (kept mapped for the life of the process: a peer may read until it exits)
        return arena

    @staticmethod
    def virtual(ops: Ops, world: int, n_total: int, state_dtypes: list, adaptive: bool) -> list:
        """`world` arenas for virtual ranks of THIS process: slices of one zeroed allocation."""
        nb = PeerArena.nbytes(ops, n_total, state_dtypes, adaptive)
        whole = torch.zeros(world * nb, dtype=torch.uint8, device=ops.device())
        return [PeerArena(whole[r * nb:(r + 1) * nb], [(o - r) * nb for o in range(world)], r, world) for r in range(world)]


class _GroupOwner:
    def __init__(self, ops, handle):
        self.ops, self.handle = ops, handle

    def __del__(self):
        try:
            self.ops.lib.call("gjx_comm_group_destroy", self.handle)
        except Exception:
            pass


class ShardedSMC:
    """Bootstrap SMC (`kind` "lgssm": BASELINE configs[2]/[3]; "hmm": configs[4]) with the population sharded
    over ranks in equal contiguous blocks of whole tiles.

    Per step: (a) ONE launch: resample + propagate + weight the rank's OWN output slots from the global previous
    population (only the source tiles that feed those slots are read), emitting the fixed-point weights and tile records
    of the new weights; (b) the all-gather of the records (three dense arrays, 144-160 B per 1024 particles; no all-reduce: they are
    anchored per tile); (c) the ancestor shuffle.  Ancestors are monotone in the output slot, so the
    sources of a rank's slots are ONE contiguous global range, known to every rank from the records alone
    (`needed_tile_ranges`):
      exchange="ranges"    each rank receives exactly that range, in place at its global offset, from the few
                           ranks that own a piece of it (grouped send/recv = an all-to-all-v over xGMI without
                           packing or a count exchange); volume ~ the rank's own block, independent of the
                           number of ranks; costs one device->host read of the ranges per step;
      exchange="allgather" every rank receives the whole population (no host sync; volume grows with ranks).
    Either way particles, ancestors and log Z are bit-identical to the single-device filter."""

    def __init__(self, ops: Ops, kind: str, impl: int, seed: int, n_total: int, T: int, rank: int, world: int,
                 record_ancestors: bool = False, exchange: str = "ranges", comm=None, poison: bool = False,
                 n_states=None, lgssm=None, y=None, plan=None, obs=None, ess_threshold: float = 0.0, arena=None):
        """`lgssm` (abi.Lgssm) / `y`: another linear-Gaussian model and observation sequence than the benchmark's.
        kind "plan": a generated filter — `plan` from `ops.smc_plan_create`, `obs` [T, n_obs]."""
        tile = ops.tile
        if n_total % (world * tile) != 0:
            raise ValueError(f"n_total must be a multiple of world*{tile}")
        if kind not in ("lgssm", "hmm", "plan") or exchange not in ("ranges", "allgather"):
            raise ValueError("kind: lgssm | hmm | plan; exchange: ranges | allgather")
        self.ops, self.kind, self.impl, self.T, self.rank, self.world = ops, kind, impl, T, rank, world
        self.exchange, self.poison = exchange, poison
        self.comm = comm if comm is not None else TorchComm(rank, world)
        self.n_total, self.n_local = n_total, n_total // world
        self.first = rank * self.n_local
        sk, rk = W.smc_key_schedule(prng.key(seed, impl), T)
        # ess_threshold in (0, 1): resample only when ESS < threshold * n_total; every rank takes the same decision
        # from the all-gathered exact ESS sums, and a step that keeps its particles exchanges nothing
        self.cfg = ops.smc_config(impl, n_total, self.first, self.n_local, sk, rk, ess_threshold)
        dev = ops.device()
        if kind == "lgssm":
            self.y = W.lgssm_data(T) if y is None else y
            self.model = W.lgssm_model() if lgssm is None else lgssm
            self.log_z_exact = W.lgssm_exact_log_z(self.y) if lgssm is None else float("nan")
            sdt = torch.float32
        elif kind == "plan":
            import numpy as np

            self.plan = plan
            self.y = np.asarray(obs, dtype=np.float32).reshape(T, -1)
            self.log_z_exact = float("nan")
            sdt = torch.float32
        else:
            trans, obs = W.hmm_tables(n_states)
            self.k = trans.shape[0]
            self.init = W.HMM["init_state"] % self.k
            self.y = W.hmm_data(T, n_states)
            self.log_z_exact = W.hmm_exact_log_z(self.y, n_states)
            self.model = ops.hmm_model(self.k, self.init, torch.from_numpy(trans).to(dev).contiguous(),
                                       torch.from_numpy(obs).to(dev).contiguous())
            self.trans_alias, self.obs_logp = ops.hmm_prepare_model(self.model)
            sdt = torch.int32
        # global-size buffers: a rank's own block is always current, remote ranges are filled on demand
        self.n_cols = plan.n_state if kind == "plan" else 1
        self.adaptive = bool(self.cfg._adaptive)
        # `arena` (PeerArena): the peer transport — the populations live in the rank's arena, at the same offsets on every rank
        self.arena = arena
        if arena is not None:
            self.pop = arena.pops(ops, n_total, [sdt] * self.n_cols, self.adaptive)
        else:
            self.pop = [ops.smc_pop(n_total, [sdt] * self.n_cols, self.adaptive) for _ in range(2)]
        for p_ in self.pop:
            for c in [*p_.state, p_.qw, p_.logw]:
                c.zero_()
        # one message per step for adaptive filters on a collective transport: [world, tiles_local, 4] int64 (records | ESS sums)
        self.stage = (torch.zeros((world, 4 * (self.n_local // tile)), dtype=torch.int64, device=dev)
                      if self.adaptive and world > 1 and arena is None else None)
        self.out_e = torch.empty(T, dtype=torch.int32, device=dev)
        self.out_q = torch.zeros(T, dtype=torch.int64, device=dev)
        self.ancestors = torch.empty((T, self.n_local), dtype=torch.int32, device=dev) if record_ancestors else None
        self.ranges = torch.zeros(2 * world + 1, dtype=torch.int64)  # host memory the device writes (pinned on a GPU box)
        if dev.type == "cuda":
            self.ranges = self.ranges.pin_memory()
        self.ranges_np, self.ticket = self.ranges.numpy(), 0
        self.received = 0  # particles received over the run (volume of the shuffle)

    def _step(self, t: int, cur: int, prv: int):
        anc = None if self.ancestors is None else self.ancestors[t]
        keep_logw = self.adaptive or t == self.T - 1
        out = self.pop[cur].struct(self.first, with_logw=keep_logw)
        prev = self.pop[prv].struct() if t else None
        pe, pq = (self.out_e[t - 1:t], self.out_q[t - 1:t]) if t else (None, None)
        if self.kind == "plan":
            self.ops.smc_plan_step(self.cfg, self.plan, t, self.y[t], prev, out, pe, pq, anc)
        elif self.kind == "lgssm":
            self.ops.smc_lgssm_step(self.cfg, self.model, t, float(self.y[t]), prev, out, pe, pq, anc)
        else:
            self.ops.smc_hmm_step(self.cfg, self.model, t, int(self.y[t]), prev, out, self.trans_alias, self.obs_logp, pe, pq, anc)

    def _shuffle(self, cur: int):
        """Make the source ranges of the next resampling present on every rank."""
        ops, tile = self.ops, self.ops.tile
        lo, hi = self.first, self.first + self.n_local
        pop = self.pop[cur]
        cols = pop.columns()
        if self.poison:  # tests: whatever is not received below must never be read
            for c in cols:
                keep = c[lo:hi].clone()
                c.fill_(float("nan") if c.dtype == torch.float32 else 0)
                c[lo:hi] = keep
        if self.exchange == "allgather":
            for c in cols:
                self.comm.all_gather(c, lo, hi)
            self.comm.all_gather(pop.subs, lo // tile, hi // tile)
            self.received += self.n_total - self.n_local
            return
        # the ranges of all ranks from the records, computed on the device and stored straight into pinned host
        # memory with a ticket behind them: the host polls for the ticket instead of synchronising the stream
        self.ticket += 1
        ops.smc_source_ranges(self.cfg, pop.recs, pop.ess, self.world, self.ranges, self.ticket)
        rh = self.ranges_np
        if self.ranges.is_pinned():
            spins = 0
            while rh[-1] != self.ticket:
                spins += 1
                if spins > 200000:  # ~tens of ms: something else holds the stream; fall back to waiting for it
                    torch.cuda.current_stream().synchronize()
                    if rh[-1] != self.ticket:
                        raise RuntimeError("gjx_smc_source_ranges did not deliver its ticket")
        ranges = rh[:-1].reshape(self.world, 2) * tile
        sends, recvs = [], []
        for j in range(self.world):
            if j == self.rank:
                continue
            a, b = max(int(ranges[j, 0]), lo), min(int(ranges[j, 1]), hi)  # what rank j needs of my block
            if a < b:
                sends.append((j, a, b))
            jl = j * self.n_local
            a, b = max(int(ranges[self.rank, 0]), jl), min(int(ranges[self.rank, 1]), jl + self.n_local)
            if a < b:
                recvs.append((j, a, b))
                self.received += b - a
        # r04: the sub-prefixes of the requested tiles travel with the shuffle (128 B per tile) instead of an all-gather of all
        self.comm.exchange(cols + [pop.subs], sends, recvs, [1] * len(cols) + [tile])

    def _result(self):
        ops = self.ops
        lo, hi = self.first, self.first + self.n_local
        last = self.pop[(self.T - 1) & 1]
        final = [c[lo:hi] for c in last.state]
        return dict(out_e=self.out_e, out_q=self.out_q, state=final[0] if self.n_cols == 1 else final,
                    logw=last.logw[lo:hi], ancestors=self.ancestors,
                    log_z=ops.log_z_from_pairs(self.out_e, self.out_q, self.n_total, self.cfg._flags),
                    resampled=self.cfg._flags, log_z_exact=self.log_z_exact, received=self.received)

    def run_native(self, comm: "NativeComm"):
        """The same filter driven from C (`gjx_smc_sharded_run_*`): the per-step launch, the all-gather of the records
        and the ancestor shuffle without the interpreter in between.  Same results as `run()` bit for bit."""
        import ctypes as C

        import numpy as np

        from . import abi

        ops = self.ops
        if self.poison and self.arena is None:  # tests: whatever a rank never receives must never be read
            for p_ in self.pop:
                for c in p_.columns(with_logw=True):
                    c.fill_(float("nan") if c.dtype == torch.float32 else -1)
        io = abi.ShardedIO()
        structs = [p_.struct() for p_ in self.pop]
        for b in range(2):
            io.pop[b] = structs[b]
        io.out_e, io.out_q = self.out_e.data_ptr(), self.out_q.data_ptr()
        io.ancestors = self.ancestors.data_ptr() if self.ancestors is not None else None
        io.ranges = self.ranges.data_ptr()
        io.shuffle = 0 if self.exchange == "ranges" else 1
        io.stage = self.stage.data_ptr() if self.stage is not None else None
        recv = C.c_uint64(0)
        io.received = C.pointer(recv)
        try:
            if self.kind == "lgssm":
                y = np.ascontiguousarray(np.asarray(self.y, dtype=np.float32))
                ops.lib.call("gjx_smc_sharded_run_lgssm", comm.handle, C.byref(self.cfg), C.byref(self.model),
                             C.c_void_p(y.ctypes.data), C.byref(io), ops.stream())
            elif self.kind == "hmm":
                y = np.ascontiguousarray(np.asarray(self.y, dtype=np.int32))
                ops.lib.call("gjx_smc_sharded_run_hmm", comm.handle, C.byref(self.cfg), C.byref(self.model),
                             C.c_void_p(y.ctypes.data), ops._p(self.trans_alias), ops._p(self.obs_logp), C.byref(io), ops.stream())
            else:
                y = np.ascontiguousarray(np.asarray(self.y, dtype=np.float32))
                ops.lib.call("gjx_smc_sharded_run_plan", comm.handle, C.byref(self.cfg), self.plan.handle,
                             C.c_void_p(y.ctypes.data) if y.size else None, C.byref(io), ops.stream())
        finally:
            self.received = int(recv.value)
        if self.arena is not None:
            self.arena.check(ops)  # a wait that timed out set the arena's error word: raise instead of returning garbage
        return self._result()

    def run(self):
        ops = self.ops
        lo, hi = self.first, self.first + self.n_local
        tl, th = lo // ops.tile, hi // ops.tile
        for t in range(self.T):
            cur, prv = t & 1, (t & 1) ^ 1
            self._step(t, cur, prv)
            pop = self.pop[cur]
            # r04: ONE all-gather per step — the records; adaptive filters pack records + ESS sums into one message
            if pop.ess is not None and self.world > 1:
                ops.smc_records_pack(self.cfg, self.world, False, pop.recs, pop.ess, self.stage)
                self.comm.all_gather(self.stage, self.rank, self.rank + 1)
                ops.smc_records_pack(self.cfg, self.world, True, pop.recs, pop.ess, self.stage)
            else:
                self.comm.all_gather(pop.recs, tl, th)
            if t + 1 < self.T:
                self._shuffle(cur)
        ops.smc_finish(self.cfg, self.pop[(self.T - 1) & 1].recs, self.out_e[self.T - 1:self.T], self.out_q[self.T - 1:self.T])
        return self._result()


def ShardedLgssmSMC(ops: Ops, impl: int, seed: int, n_total: int, T: int, rank: int, world: int,
                    record_ancestors: bool = False, **kw):
    return ShardedSMC(ops, "lgssm", impl, seed, n_total, T, rank, world, record_ancestors, **kw)

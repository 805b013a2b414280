"""x-slab decomposition of one problem over the GPUs of a node: one process per GPU.

The reference is single-process (README.md:60-61); this module is new.  The grid is cut along x
(the slow axis of the device planes, so halo rows are contiguous).  A fused step needs its
neighbours' field only one row deep (the composed MacCormack stencil has radius 1, see
csrc/step_kernel.hip), so per time step each rank needs ONE row from each neighbour plus everybody's 64-byte record
(sum Ekin, max v^2, max c^2, validity).  Both travel in a single all-gather after the stencil
kernel, before the dt/residual commit:

    gpf_step_local   (stage-1 ghost data + fused stencil + local ghost rules + message: first row, last row, record)
    all_gather_into_tensor(gathered, message)               <- torch.distributed (RCCL over xGMI)
    gpf_step_commit  (scatter the neighbours' rows, reduce records in rank order, advance dt/residual)

Everything is enqueued on one stream; there is no host synchronisation inside `advance`.
Reductions are combined in rank order on every rank, so all ranks hold bit-identical dt.

Kinds of a slab's outer rows (gpf_config.halo_lo/hi): 0 physical ghost row (local rule),
1 copy of the neighbour's interior row, 2 the domain's periodic ghost row (filled by the ring).

`SlabDriver` is backend-neutral (any engine with message / step_local / commit); the product
engine is `HipSlabEngine`.  tests/test_slab_gloo.py drives the same driver with a CPU engine over
gloo to check partitioning, message pairing and the rank-ordered reduction.
"""
import ctypes as C
import io as _io
import os
import sys
from copy import deepcopy

import numpy as np

from . import _lib
from .io import read_yaml_input
from .topography import Topography, topography_rows

HALO_PHYSICAL, HALO_NEIGHBOUR, HALO_SEAM = 0, 1, 2


def partition(Nx, world):
    """Global interior rows [lo, hi] (1-based, inclusive) of every rank; remainders go to the first ranks."""
    base, rem = divmod(Nx, world)
    if base < 1:
        raise ValueError(f"cannot cut Nx={Nx} rows into {world} slabs")
    out, lo = [], 1
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((lo, lo + n - 1))
        lo += n
    return out


class SlabLayout:
    """Everything a rank needs to know about its slab."""

    def __init__(self, grid, rank, world):
        self.rank, self.world = rank, world
        self.Nx_global = grid['Nx']
        self.lo, self.hi = partition(grid['Nx'], world)[rank]
        self.nx = self.hi - self.lo + 1
        periodic = all(grid['bc_xE_P']) and all(grid['bc_xW_P'])
        self.periodic = periodic
        first, last = rank == 0, rank == world - 1
        if world == 1:
            self.kind_lo = self.kind_hi = HALO_PHYSICAL
            self.lower = self.upper = None
        else:
            self.kind_lo = (HALO_SEAM if periodic else HALO_PHYSICAL) if first else HALO_NEIGHBOUR
            self.kind_hi = (HALO_SEAM if periodic else HALO_PHYSICAL) if last else HALO_NEIGHBOUR
            self.lower = (world - 1 if periodic else None) if first else rank - 1
            self.upper = (0 if periodic else None) if last else rank + 1

    def rows(self):
        """Slice of the global (Nx+2)-row arrays held by this rank, including its two outer rows."""
        return slice(self.lo - 1, self.hi + 2)

    def local_grid(self, grid):
        g = deepcopy(grid)
        g['Nx'] = self.nx
        g['Lx'] = grid['dx'] * self.nx
        return g


class LoopbackGroup:
    """One process standing in for rank `rank` of `world` (rehearsal and timing on a one-GPU box): every peer's contribution
    to a collective is this rank's own.  For a domain that repeats with the slab's period that is exactly what the real
    neighbours would send (tests/test_gpu_gp_large.py); for any other domain it exercises the rank's kernels, message
    sizes and launch sequence, not the physics across the slab boundary."""

    class ReduceOp:
        SUM, MAX, MIN = 'sum', 'max', 'min'

    def __init__(self, rank, world):
        self._rank, self._world = rank, world

    def get_rank(self):
        return self._rank

    def get_world_size(self):
        return self._world

    def all_gather_into_tensor(self, out, inp):
        out.view(self._world, -1).copy_(inp.view(1, -1).expand(self._world, -1))

    def all_reduce(self, t, op=None):
        if op == self.ReduceOp.SUM:
            t.mul_(self._world)

    def broadcast_object_list(self, box, src=0):
        pass

    def barrier(self):
        pass


class HostStagedGroup:
    """torch.distributed look-alike that moves GPU tensors through host memory over a CPU backend (gloo).

    RCCL refuses two ranks on one device ("Duplicate GPU detected"), so on a one-GPU box the multi-rank behaviour of the
    HIP slab engine (halo kinds, seam topography, message rows, rank-ordered commit) is rehearsed with a few processes
    sharing the GPU and this transport (tests/test_gpu_slab.py, bench.py with GPF_BENCH_ONE_GPU_REHEARSAL=1)."""

    def __init__(self, dist, torch):
        self._d, self._t = dist, torch
        self.ReduceOp = dist.ReduceOp

    def get_rank(self):
        return self._d.get_rank()

    def get_world_size(self):
        return self._d.get_world_size()

    def all_gather_into_tensor(self, out, inp):
        self._t.cuda.synchronize()
        o, i = out.cpu(), inp.cpu()
        self._d.all_gather_into_tensor(o, i)
        out.copy_(o)

    def all_reduce(self, t, op=None):
        c = t.cpu()
        self._d.all_reduce(c, op=op if op is not None else self.ReduceOp.SUM)
        t.copy_(c)

    def broadcast_object_list(self, box, src=0):
        self._d.broadcast_object_list(box, src=src)

    def barrier(self):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()


class ThreadWorld:
    """N ranks as N THREADS of one process, all on one GPU: `run(fn)` calls fn(group) once per rank, each with its own
    torch.distributed look-alike whose collectives meet at thread barriers and move data with device copies.

    What it is for: the test boxes allow six processes on a card, so an 8-rank decomposition (BASELINE.json configs[4], the
    8-slab strong-scaling run) cannot be rehearsed with one process per rank there.  Every rank is a real SlabProblem with
    its own library handle, partition, halo kinds, seam topography and messages; only the transport is stood in for.  All
    ranks enqueue on the device's default stream, so "every rank has enqueued its part" (a thread barrier) is all the
    ordering a collective needs.  All-gather transport only: the mailbox kernels poll for their peers and would wait for
    launches queued behind them on the shared stream."""

    class ReduceOp:
        SUM, MAX, MIN = 'sum', 'max', 'min'

    def __init__(self, world, torch):
        import threading
        self.world, self.torch = world, torch
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world

    class _Group:
        def __init__(self, w, rank):
            self._w, self._rank = w, rank
            self.ReduceOp = ThreadWorld.ReduceOp

        def get_rank(self):
            return self._rank

        def get_world_size(self):
            return self._w.world

        def barrier(self):
            self._w._barrier.wait()

        def all_gather_into_tensor(self, out, inp):
            w = self._w
            w._slots[self._rank] = inp
            w._barrier.wait()                           # every rank's producers are enqueued
            parts = out.view(w.world, -1)
            for r in range(w.world):
                parts[r].copy_(w._slots[r].view(-1))
            w._barrier.wait()                           # every rank's copies are enqueued: the inputs may change again

        def all_reduce(self, t, op=None):
            w, torch = self._w, self._w.torch
            w._slots[self._rank] = t.clone()
            w._barrier.wait()
            stack = torch.stack([w._slots[r] for r in range(w.world)])      # rank order on every rank
            res = stack.sum(0) if op in (None, 'sum') else (stack.max(0).values if op == 'max' else stack.min(0).values)
            w._barrier.wait()
            t.copy_(res)

        def broadcast_object_list(self, box, src=0):
            w = self._w
            if self._rank == src:
                w._slots[src] = list(box)
            w._barrier.wait()
            box[:] = w._slots[src]
            w._barrier.wait()

    def run(self, fn):
        """fn(group) on every rank; returns the results in rank order.  The first exception breaks the barriers (the other
        ranks then fail at their next collective) and is re-raised."""
        import threading
        results, errors = [None] * self.world, []

        def body(rank):
            try:
                if self.torch.cuda.is_available():
                    self.torch.cuda.set_device(0)
                results[rank] = fn(ThreadWorld._Group(self, rank))
            except threading.BrokenBarrierError:
                pass
            except BaseException as e:      # noqa: BLE001
                errors.append((rank, e))
                self._barrier.abort()

        threads = [threading.Thread(target=body, args=(r,), name=f'slab-rank-{r}') for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            rank, e = sorted(errors, key=lambda x: x[0])[0]
            raise RuntimeError(f"rank {rank} of the thread world failed: {type(e).__name__}: {e}") from e
        return results


class SlabDriver:
    """Runs the split step of an engine: step_local -> ONE all-gather -> commit.

    Every rank contributes one message [first row | last row | record]; after the all-gather each rank picks its
    neighbours' rows out of the gathered buffer.  (Point-to-point sends would move 8x fewer bytes, but the rows are
    ~100 KB and a step is latency-bound: one collective instead of an all-gather plus a send/recv group halves both
    the host dispatch and the number of RCCL kernels per step.)"""

    def __init__(self, engine, layout, dist, torch):
        self.engine, self.layout, self.dist = engine, layout, dist
        self.message = engine.message()
        self.gathered = torch.zeros(self.message.numel() * layout.world, dtype=torch.float64, device=self.message.device)
        # rank whose LAST row is my row 0 / whose FIRST row is my row Nx+1
        self.rank_lo = -1 if layout.lower is None else layout.lower
        self.rank_hi = -1 if layout.upper is None else layout.upper

        self.torch = torch
        self.enqueued = 0               # steps issued since pre_run (the predictor direction may alternate with it)
        self.graph = None
        self.graph_honor = None
        self.use_graph = os.environ.get('GPF_SLAB_GRAPH', '0') == '1' and self.message.is_cuda
        self.p2p = False                # True: the step's own kernels exchange rows and records (gpf_step_p2p)

    def _eager(self, n, honor_stop):
        for _ in range(n):
            self.engine.step_local(honor_stop)
            self.dist.all_gather_into_tensor(self.gathered, self.message)
            self.engine.commit(self.gathered, honor_stop, self.rank_lo, self.rank_hi)
        self.enqueued += n

    def _capture(self, honor_stop):
        """Record two consecutive steps (an even and an odd one: MC_order 0 alternates the sweep direction,
        problem.py:521-522) -- library kernels and the RCCL all-gather -- into one hipGraph."""
        torch = self.torch
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.engine.set_stream(side.cuda_stream)
            try:
                with torch.cuda.graph(g, stream=side):
                    for _ in range(2):
                        self.engine.step_local(honor_stop)
                        self.dist.all_gather_into_tensor(self.gathered, self.message)
                        self.engine.commit(self.gathered, honor_stop, self.rank_lo, self.rank_hi)
            finally:
                self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.graph, self.graph_honor = g, honor_stop

    def advance(self, n, honor_stop=False):
        """n time steps; with GPF_SLAB_GRAPH=1 pairs of steps are replayed from a captured hipGraph, which takes
        the per-step host dispatch (two library calls and one collective) off the critical path."""
        if self.p2p:
            self.engine.step_p2p(n, honor_stop)
            self.enqueued += n
            return
        if not self.use_graph or n < 6:
            return self._eager(n, honor_stop)
        if self.enqueued == 0:
            self._eager(2, honor_stop)          # first calls allocate and plan: keep them out of the capture
            n -= 2
        if self.enqueued % 2:
            self._eager(1, honor_stop)
            n -= 1
        if self.graph is None or self.graph_honor != honor_stop:
            self._capture(honor_stop)           # capturing does not execute
        for _ in range(n // 2):
            self.graph.replay()
        self.enqueued += 2 * (n // 2)
        self.engine.resync()                    # the library's own step counter did not see the replays
        self._eager(n % 2, honor_stop)


class _DeviceArray:
    """Exposes library-owned device memory to torch through the CUDA array interface (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f8', 'data': (int(ptr), False), 'version': 2}


class HipSlabEngine:
    """The product engine: one libgapflow_hip handle per rank."""

    def __init__(self, lib, handle, torch, world):
        self.lib, self.h, self.torch, self.world = lib, handle, torch, world
        p, n = C.c_void_p(), C.c_size_t(0)
        _lib.check(lib.gpf_slab_message(handle, C.byref(p), C.byref(n)))
        self._msg = torch.as_tensor(_DeviceArray(p.value, n.value), device='cuda')

    def message(self):
        return self._msg

    def step_local(self, honor_stop):
        _lib.check(self.lib.gpf_step_local(self.h, int(honor_stop)))

    def commit(self, gathered, honor_stop, rank_lo, rank_hi):
        _lib.check(self.lib.gpf_step_commit(self.h, int(honor_stop), C.c_void_p(gathered.data_ptr()), self.world,
                                            int(rank_lo), int(rank_hi)))

    def step_p2p(self, n, honor_stop):
        _lib.check(self.lib.gpf_step_p2p(self.h, int(n), int(honor_stop)))

    def set_stream(self, stream_ptr):
        _lib.check(self.lib.gpf_set_stream(self.h, C.c_void_p(stream_ptr)))

    def resync(self):
        sc = _lib.GpfScalars()
        _lib.check(self.lib.gpf_state(self.h, C.byref(sc)))     # synchronises; re-reads the device step counter


class _DomainFeatures:
    """The whole domain's (ncell, 7) feature table [rho, jx, jy, h, dh/dx, dh/dy, extra] of the initial state, rows on
    demand: what Database.initialize asks of its `Xtest` (column means of the uniform initial field, the number of
    cells, the features of a few hundred sampled cells) without a 3.8 GB table per rank at 8192^2."""

    def __init__(self, q0, grid, topo_rows):
        self._q0, self._topo_rows = np.asarray(q0, float), topo_rows
        self._ny2 = grid['Ny'] + 2
        self.shape = ((grid['Nx'] + 2) * self._ny2, 7)

    def column_mean(self, k):
        if k > 2:
            raise NotImplementedError("only the (uniform) initial field has a mean here")
        return float(self._q0[k])

    def rows(self, cells):
        cells = np.asarray(cells, int)
        ix, iy = cells // self._ny2, cells % self._ny2
        uniq, inv = np.unique(ix, return_inverse=True)
        t = self._topo_rows(uniq)                           # (3, len(uniq), Ny+2)
        out = np.zeros((len(cells), 7))
        out[:, :3] = self._q0
        out[:, 3:6] = t[:, inv, iy].T
        return out


class SlabProblem:
    """A Problem cut into x-slabs; construct it on every rank of an initialised process group."""

    def __init__(self, input_dict, device=0, dist=None):
        import torch
        if dist is None:
            import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self._device, self._writer_problem, self._pre_run_done = device, None, False
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.input = input_dict
        grid, prop, geo = input_dict['grid'], input_dict['properties'], input_dict['geometry']
        self.layout = L = SlabLayout(grid, self.rank, self.world)
        self.lib = _lib.require_device()

        # the serial problem's configuration, narrowed to this slab
        from .problem import Problem
        shell = Problem.__new__(Problem)
        shell.options, shell.numerics, shell.prop, shell.geo = input_dict['options'], input_dict['numerics'], prop, geo
        shell.grid = L.local_grid(grid)
        cfg = shell._make_config(device)
        # shear thinning: the viscosity of a halo row depends on grad p there, i.e. on a second row of the neighbour -- the
        # stage-wise step then exchanges two rows per side (gpf_upload_beyond / gpf_stage_message) and is the only step
        self._thinning = bool(cfg.thinning)
        if self._thinning and L.nx < 2 and self.world > 1:
            raise ValueError("shear thinning across slabs needs at least two rows per slab")
        cfg.halo_lo, cfg.halo_hi = L.kind_lo, L.kind_hi
        self.grid_local = shell.grid
        self._h = C.c_void_p()
        _lib.check(self.lib.gpf_create(C.byref(cfg), C.byref(self._h)))
        _lib.check(self.lib.gpf_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

        # this slab's rows of the domain's topography (never the whole table: 2 GB per rank at 8192^2)
        hmins = self._shared_asperity_heights(geo)
        if geo.get('flip'):
            whole = Topography(grid, geo, prop).full[:3]
            topo_rows = lambda rows: whole[:, np.asarray(rows)]
        else:
            topo_rows = lambda rows: topography_rows(grid, geo, rows, hmins)
        self._topo_rows = topo_rows
        shape = (L.nx + 2, grid['Ny'] + 2)
        self._shape = shape
        topo_local = topo_rows(np.arange(L.lo - 1, L.hi + 2))
        self._upload(_lib.FIELD_TOPO, topo_local)
        q = np.empty((3,) + shape)
        q[0], q[1], q[2] = prop['rho0'], prop['rho0'] * geo['U'] / 2.0, prop['rho0'] * geo['V'] / 2.0
        self._upload(_lib.FIELD_Q, q)
        if self._thinning:
            beyond = _lib.f64c(np.full(grid['Ny'] + 2, prop['rho0']))          # the uniform initial field, one row further
            for side, kind in ((0, L.kind_lo), (1, L.kind_hi)):
                if kind == HALO_NEIGHBOUR:
                    _lib.check(self.lib.gpf_upload_beyond(self._h, side, _lib.as_dp(beyond), beyond.size))
        nxg = grid['Nx']
        for side, kind, (r_src, r_up) in ((0, L.kind_lo, (nxg, nxg + 1)), (1, L.kind_hi, (1, 0))):
            if kind == HALO_SEAM:
                seam = np.zeros((2, 4, grid['Ny'] + 2))
                two = topo_rows([r_src, r_up])
                seam[0, :3], seam[1, :3] = two[:, 0], two[:, 1]
                seam = _lib.f64c(seam)
                _lib.check(self.lib.gpf_set_seam_topo(self._h, side, _lib.as_dp(seam), seam.size))
        self.engine = HipSlabEngine(self.lib, self._h, torch, self.world)
        self.driver = SlabDriver(self.engine, L, dist, torch)

        # surrogate closures: the database and the models are replicated on every rank
        self.grid, self._lib, self._closures_stale = grid, self.lib, True
        self._topo_local = topo_local
        self._extra = np.zeros((1,) + shape)
        self._features_global = _DomainFeatures(q[:, 0, 0], grid, topo_rows)
        self.database, self._gp_models = None, {}
        gp = input_dict.get('gp')
        if input_dict.get('db') is not None:
            from .gp import make_database
            self.database = make_database(input_dict, device)
        if gp is not None:
            if self.database is None:
                raise IOError("a `gp` section needs a `db` section")
            from .gp import attach_surrogates, SlabSurrogate
            self._gp_models = attach_surrogates(self, gp, self.database, cls=SlabSurrogate)
            self._pair = torch.zeros(8, dtype=torch.float64, device='cuda')
            self._pairs = torch.zeros(8 * self.world, dtype=torch.float64, device='cuda')

        if os.environ.get('GPF_SLAB_TRANSPORT', 'rccl') == 'p2p':
            if not self.connect_p2p():
                raise RuntimeError("GPF_SLAB_TRANSPORT=p2p: the peers' mailboxes could not be mapped (HIP IPC)")

    def _shared_asperity_heights(self, geo):
        """topography.py:141-146 draws the minimum heights of num^2 asperities from an unseeded normal distribution: every
        rank must use rank 0's draw."""
        if geo['type'] != 'asperity' or geo['num'] == 1:
            return None
        h0, h1, num = geo['hmin'], geo['hmax'], geo['num']
        box = [np.random.normal(loc=h0 + (h1 - h0) / 2., scale=(h1 - h0) / 2. / 2.57, size=num**2) if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def connect_p2p(self):
        """Switch the fused slab step to the peer-to-peer transport (ranks of one node): exchange the mailboxes'
        IPC handles once through the process group, map them, and from then on `advance` enqueues whole batches of
        steps whose kernels talk to each other directly.  Collective; returns False (and stays on the all-gather
        transport) unless every rank succeeded."""
        t, L = self.torch, self.layout
        buf = (C.c_ubyte * 64)()
        ok = self.lib.gpf_p2p_export(self._h, buf, 64) == 0
        mine = t.tensor(list(buf), dtype=t.uint8).to('cuda')
        everyone = t.zeros(64 * self.world, dtype=t.uint8, device='cuda')
        self.dist.all_gather_into_tensor(everyone, mine)
        if ok:
            handles = everyone.cpu().numpy().tobytes()
            d = self.driver
            ok = self.lib.gpf_p2p_connect(self._h, self.rank, self.world, handles, d.rank_lo, d.rank_hi) == 0
        if not ok:
            print(f"[gapflow_amd] rank {self.rank}: peer-to-peer transport unavailable: "
                  f"{self.lib.gpf_last_error().decode()}", file=sys.stderr)
        flag = t.tensor([1.0 if ok else 0.0], dtype=t.float64, device='cuda')
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        self.driver.p2p = bool(flag.item() == 1.0)
        return self.driver.p2p

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self.lib.gpf_destroy(h)
            h.value = None

    @classmethod
    def from_string(cls, text, device=0, dist=None):
        with _io.StringIO(text) as f:
            return cls(read_yaml_input(f), device=device, dist=dist)

    @classmethod
    def from_yaml(cls, fname, device=0):
        with open(fname) as f:
            return cls(read_yaml_input(f), device=device)

    def _upload(self, field, arr):
        a = _lib.f64c(arr)
        _lib.check(self.lib.gpf_upload(self._h, field, _lib.as_dp(a), a.size))

    def global_scalars(self):
        """Ekin, v_max, v_sound, mass over the whole domain (sum / max over the slabs)."""
        sc = _lib.GpfScalars()
        _lib.check(self.lib.gpf_scalars(self._h, C.byref(sc)))
        t = self.torch
        s = t.tensor([sc.ekin, sc.mass], dtype=t.float64, device='cuda')
        m = t.tensor([sc.v_max, sc.v_sound], dtype=t.float64, device='cuda')
        self.dist.all_reduce(s, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX)
        return {'ekin': float(s[0]), 'mass': float(s[1]), 'v_max': float(m[0]), 'v_sound': float(m[1])}

    # -- what a SlabSurrogate asks of its problem -----------------------------------------------
    def _download(self, field, ncomp):
        out = np.empty((ncomp,) + self._shape)
        _lib.check(self.lib.gpf_download(self._h, field, _lib.as_dp(out), out.size))
        return out

    def _sync_to_device(self):
        pass

    def _gather_pairs(self, value, payload):
        t = self.torch
        self._pair.copy_(t.tensor([value] + list(payload), dtype=t.float64))
        self.dist.all_gather_into_tensor(self._pairs, self._pair)
        return self._pairs.cpu().numpy().reshape(self.world, 8)

    def domain_max(self, value):
        return float(np.max(self._gather_pairs(value, np.zeros(7))[:, 0]))

    def features_of_domain_max(self, value, features):
        """Feature row of the most uncertain cell of the whole domain: the lowest rank holding the maximum wins,
        like the first hit of the serial argmax (rows shared by two slabs carry identical features)."""
        pairs = self._gather_pairs(value, features)
        return pairs[int(np.argmax(pairs[:, 0])), 1:].copy()

    def _features_local(self):
        return np.vstack([self.local_q(), self._topo_local, self._extra]).reshape(7, -1).T

    def pre_run(self):
        """Problem._pre_run (problem.py:412-443) with domain-wide scalars."""
        if self._gp_models:
            self.database.initialize(self._features_global, self.grid['dim'])       # identical on every rank
            for m in self._gp_models.values():
                m.init()
        _lib.check(self.lib.gpf_pre_run(self._h))
        g = self.global_scalars()
        num, grid = self.input['numerics'], self.input['grid']
        dt_crit = min(grid['dx'], grid['dy']) / (g['v_max'] + g['v_sound'])
        dt = num['CFL'] * dt_crit if num['adaptive'] else num['dt']
        _lib.check(self.lib.gpf_set_dt(self._h, float(dt)))
        _lib.check(self.lib.gpf_set_ekin_old(self._h, float(g['ekin'])))
        self._pre_run_done = True

    def advance(self, n, honor_stop=False, write_freq=None):
        if not self._gp_models and not self._thinning:
            return self.driver.advance(n, honor_stop)
        for _ in range(n):
            st = self.state()
            if st.invalid or (honor_stop and (st.converged or st.step >= self.input['numerics']['max_it'])):
                break
            self._stagewise_step(write_freq or self.input['options']['write_freq'], st.step)

    def _stagewise_step(self, write_freq, step):
        """Problem._update_with_surrogates for a slab: the working field's outer rows travel after each stage
        (the closures of the next stage read them), the scalar records once at the end."""
        lib, h, d = self.lib, self._h, self.driver
        gathered = C.c_void_p(d.gathered.data_ptr())
        one_step_before_output = (step + 1) % write_freq == 0
        feats = None

        def features_of_cell(i):
            nonlocal feats
            if feats is None:
                feats = self._features_local()
            return feats[i]

        _lib.check(lib.gpf_open_step(h))
        for i in range(2):
            for m in self._gp_models.values():
                m.sync_scales()
            _lib.check(lib.gpf_stage_closures(h))
            changed = False
            for name in ('zz', 'xz', 'yz'):
                m = self._gp_models.get(name)
                if m is not None:
                    changed |= m.stage(i == 0, one_step_before_output, features_of_cell)
            if changed:
                _lib.check(lib.gpf_stage_closures(h))
            _lib.check(lib.gpf_stage_advance(h, i))
            _lib.check(lib.gpf_stage_message(h))
            self.dist.all_gather_into_tensor(d.gathered, d.message)
            _lib.check(lib.gpf_stage_absorb(h, gathered, self.world, d.rank_lo, d.rank_hi))
        for m in self._gp_models.values():
            m.sync_scales()
        _lib.check(lib.gpf_close_step_local(h))
        self.dist.all_gather_into_tensor(d.gathered, d.message)
        sc = _lib.GpfScalars()
        _lib.check(lib.gpf_close_step_commit(h, gathered, self.world, C.byref(sc)))
        return sc

    def state(self):
        sc = _lib.GpfScalars()
        _lib.check(self.lib.gpf_state(self._h, C.byref(sc)))
        if sc.invalid == 3:
            raise RuntimeError("slab step: a peer rank did not deliver its rows within 30 s (peer-to-peer transport)")
        return sc

    # -- Problem.run for slabs -----------------------------------------------------------------
    def gather_q(self):
        """The whole field (3, Nx+2, Ny+2) on every rank: owned rows from their owners, the domain's two ghost rows
        from the first and last rank.  Collective; meant for output frames, not for the time loop."""
        t, L = self.torch, self.layout
        parts = partition(L.Nx_global, self.world)
        max_nx = max(hi - lo + 1 for lo, hi in parts)
        ny2 = self._shape[1]
        mine = np.zeros((3, max_nx + 2, ny2))
        mine[:, :L.nx + 2] = self.local_q()
        send = t.from_numpy(mine.reshape(-1)).to('cuda')
        recv = t.zeros(send.numel() * self.world, dtype=t.float64, device='cuda')
        self.dist.all_gather_into_tensor(recv, send)
        allq = recv.cpu().numpy().reshape(self.world, 3, max_nx + 2, ny2)
        out = np.empty((3, L.Nx_global + 2, ny2))
        for r, (lo, hi) in enumerate(parts):
            out[:, lo:hi + 1] = allq[r, :, 1:hi - lo + 2]
        out[:, 0] = allq[0, :, 0]
        out[:, -1] = allq[-1, :, parts[-1][1] - parts[-1][0] + 2]
        return out

    def _frame(self):
        """One output frame through a whole-domain Problem that only rank 0 holds: it owns the output directory, the
        NetCDF / csv writers and the closures of the frame (Problem.write, problem.py:616-637)."""
        q = self.gather_q()
        if self.rank != 0:
            return
        if self._writer_problem is None:
            from .problem import Problem
            d = deepcopy(self.input)
            # with surrogates the writer shares this rank's database and mirrors its models (same data, same
            # hyper-parameters), so the closures of a frame are the surrogates' means as in a serial run
            self._writer_problem = Problem(d['options'], d['grid'], d['numerics'], d['properties'], d['geometry'],
                                           gp=d.get('gp') if self._gp_models else None,
                                           database=self.database if self._gp_models else None, device=self._device)
            w = self._writer_problem
            w.history = {k: [] for k in ('step', 'time', 'ekin', 'residual', 'vsound')}
        w = self._writer_problem
        for name, m in self._gp_models.items():
            wm = w._gp_models[name]
            if wm.theta is None or wm.last_fit_train_size != m.last_fit_train_size or not np.array_equal(wm.theta, m.theta):
                wm.theta, wm.last_fit_train_size = np.array(m.theta), m.last_fit_train_size
                wm.attach()
        st = self.state()
        w.q[...] = q
        w.step, w.simtime, w.dt, w.residual = int(st.step), st.simtime, st.dt, st.residual
        w.write(params=False)

    def run(self):
        """Problem.run (problem.py:368-410) for a slab-decomposed problem: same stopping rules (tolerance on the last
        five residuals, max_it, invalid state), same status lines and output files, written by rank 0."""
        import datetime as _dt
        opt, num = self.input['options'], self.input['numerics']
        silent, wf, max_it = opt['silent'], opt['write_freq'], num['max_it']
        if not self._pre_run_done:
            self.pre_run()
        if not silent:
            if self.rank == 0:
                print(61 * '-')
                print(f"{'Step':6s} {'Timestep':10s} {'Time':10s} {'CFL':10s} {'Residual':10s}")
                print(61 * '-')
            self._frame()
        tic = _dt.datetime.now()
        st = self.state()
        while not st.converged and st.step < max_it and not st.invalid:
            n = min(wf - st.step % wf, max_it - st.step, 4096)
            self.advance(n, honor_stop=True)
            st = self.state()
            if st.invalid:
                if self.rank == 0:
                    print('NaN detected.' if st.invalid == 1 else 'Negative density detected.',
                          'Writing previous step and aborting simulation.')
                break
            if st.step % wf == 0 and not silent:
                self._frame()
        if not silent and st.step % wf != 0:
            self._frame()
        if self.rank == 0:
            wall = _dt.datetime.now() - tic
            print(33 * '=')
            print("Total walltime   : ", str(wall).split('.')[0])
            print(f"({st.step / max(wall.total_seconds(), 1e-12):.2f} steps/s)")
            print(33 * '=')
            if not silent:
                from .io import history_to_csv
                w = self._writer_problem
                w._writer.close()
                history_to_csv(os.path.join(w.outdir, 'history.csv'), w.history)
        return st

    def local_q(self):
        out = np.empty((3,) + self._shape)
        _lib.check(self.lib.gpf_download(self._h, _lib.FIELD_Q, _lib.as_dp(out), out.size))
        return out

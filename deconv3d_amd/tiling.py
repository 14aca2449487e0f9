# coding=utf-8
"""
Spatial tiling of ONE chain over several GPUs (BASELINE config 4).

The reference is a single process; nothing here has a counterpart in it.  Two of
its properties make the tiling possible: an update at spaxel (y, x) touches only
its FSF window (lib/run.py:404-419), and the scan order of a sweep is declared
overridable (lib/run.py:553-560).

Design (MI355X: few, large point-to-point transfers over xGMI instead of one
small exchange per colour class):

* the spaxel grid is cut into ty x tx TILES of owned spaxels; a rank stores its
  tile plus a frame of two FSF half widths (the REGION; the cells its windows
  touch -- tile plus one half width -- are the USED cells);
* every tile is cut into up to four PARTS by its distance to the tile's lower /
  right border: FF (far from both), FN (within 2 half widths of the right
  border), NF (of the lower border), NN (of both).  A sweep runs four PHASES --
  all FF parts, all FN, all NF, all NN -- and within a phase every part runs
  its fh*fw colour launches.  Parts of one phase on different tiles are at least
  one full window apart, so their windows are disjoint: a phase runs on all GPUs
  at once WITHOUT communication, and the sweep equals a sequential scan in the
  order (phase, colour, spaxel) -- a single device given the same parts
  (`apply_parts`) produces the same chain bit for bit;
* after a phase each rank sends the residual cells its parts touched that another
  rank uses: one rectangle per neighbour (row strips: 2*fhh rows x W x D doubles
  = 3 MB at 300x300x128, contiguous), received as a plain copy.  Two to four bulk
  copies per sweep replace the fh*fw = 121 dependent exchanges of a per-colour
  protocol; random numbers and colour classes are keyed by GLOBAL coordinates.

Transports: RCCL inside the library (`Engine.comm_init` + `d3d_mh_sweeps`:
pack -> ncclSend/ncclRecv -> unpack on the context's stream, no host hop);
device-to-device copies between contexts of one process (`sweep_loopback`, the
one-GPU rehearsal and the bit-identity tests); host-staged torch.distributed
(gloo) for multi-process rehearsal on a box with fewer GPUs than ranks.

What this buys -- measured, DESIGN.md section 7: the colour classes of a part are a
dependent chain of fh*fw launches, and a launch that does not fill the chip costs
~12 us at 128 channels however few windows it holds (k_mh_small, round 4; 14 us in
round 3).  A sweep's compute time is the sum over the PHASES of the slowest rank in each
(`critical_path`): with 121 launches per phase and part that is >= 121 x 12 us x phases --
2.9 ms for row strips (two phases), ~5.8 ms for a 2-D grid (four) -- against 4.9 ms for the
whole 300x300x128 cube on one GPU.  The tiled chain is the mode for a cube or a chain that
must be SPLIT, or one much larger than 300x300, where every part fills a chip; it does not
make a 300x300 chain much faster.  Row strips (N x 1: two phases) are the default layout;
2-D grids (four phases) are supported and exact.  The throughput mode across GPUs is the
ensemble.

Engines are duck-typed (`mh_phase`, `halo_pack/unpack`, `halo_download/upload`,
...): the product engine is `_lib.Engine`; the CPU tests drive the same code with
an oracle-backed engine.
"""
from __future__ import annotations

import numpy as np

PLAN_PARAMS = 16     # _lib.PLAN_PARAMS / D3D_PLAN_PARAMS


def _splits(n, parts):
    """Boundaries of `parts` nearly equal consecutive ranges covering [0, n)."""
    base, extra = divmod(n, parts)
    edges = [0]
    for i in range(parts):
        edges.append(edges[-1] + base + (1 if i < extra else 0))
    return edges


def _intersect(a, b):
    """Intersection of two rectangles (y0, y1, x0, x1), or None when empty."""
    if a is None or b is None:
        return None
    y0, y1, x0, x1 = max(a[0], b[0]), min(a[1], b[1]), max(a[2], b[2]), min(a[3], b[3])
    if y0 >= y1 or x0 >= x1:
        return None
    return (y0, y1, x0, x1)


class TileLayout(object):
    """Geometry of a ty x tx tiling of an H x W spaxel grid for an fh x fw FSF."""

    def __init__(self, H, W, fh, fw, ty, tx):
        if ty < 1 or tx < 1 or ty > H or tx > W:
            raise ValueError("cannot cut %dx%d spaxels into %dx%d tiles" % (H, W, ty, tx))
        self.H, self.W, self.fh, self.fw, self.ty, self.tx = H, W, fh, fw, ty, tx
        self.fhh, self.fhw = (fh - 1) // 2, (fw - 1) // 2
        self.row_edges = _splits(H, ty)
        self.col_edges = _splits(W, tx)
        self.n = ty * tx
        # parts of one phase on neighbouring tiles must be a full window apart
        if ty > 1 and min(np.diff(self.row_edges)) < 4 * self.fhh:
            raise ValueError("tiles of %d rows are too short for an FSF of %d rows (need >= %d)"
                             % (min(np.diff(self.row_edges)), fh, 4 * self.fhh))
        if tx > 1 and min(np.diff(self.col_edges)) < 4 * self.fhw:
            raise ValueError("tiles of %d columns are too narrow for an FSF of %d columns "
                             "(need >= %d)" % (min(np.diff(self.col_edges)), fw, 4 * self.fhw))

    # -- rectangles (global coordinates, (y0, y1, x0, x1)) -------------------------
    def owned(self, rank):
        """Spaxels rank updates."""
        iy, ix = divmod(rank, self.tx)
        return (self.row_edges[iy], self.row_edges[iy + 1],
                self.col_edges[ix], self.col_edges[ix + 1])

    def _grown(self, rect, k):
        return (max(rect[0] - k * self.fhh, 0), min(rect[1] + k * self.fhh, self.H),
                max(rect[2] - k * self.fhw, 0), min(rect[3] + k * self.fhw, self.W))

    def used(self, rank):
        """Cells rank's windows touch: owned + one FSF half width."""
        return self._grown(self.owned(rank), 1)

    def region(self, rank):
        """Cells rank stores: owned + two half widths, so that a residual rebuilt from
        the region's parameters (lib/run.py:521-534) is exact on the used cells."""
        return self._grown(self.owned(rank), 2)

    def parts(self, rank):
        """[(phase, rectangle)] of rank's non-empty parts: phase 0 FF, 1 FN, 2 NF, 3 NN."""
        iy, ix = divmod(rank, self.tx)
        y0, y1, x0, x1 = self.owned(rank)
        my = 2 * self.fhh if iy < self.ty - 1 else 0     # a neighbour below
        mx = 2 * self.fhw if ix < self.tx - 1 else 0     # a neighbour to the right
        ys, xs = max(y0, y1 - my), max(x0, x1 - mx)
        cand = [(0, (y0, ys, x0, xs)), (1, (y0, ys, xs, x1)),
                (2, (ys, y1, x0, xs)), (3, (ys, y1, xs, x1))]
        return [(ph, r) for ph, r in cand if r[0] < r[1] and r[2] < r[3]]

    @property
    def phases(self):
        """Phase numbers in use, in order."""
        return sorted({ph for rank in range(self.n) for ph, _ in self.parts(rank)})

    def all_parts(self):
        """Every rank's parts: given to ONE device (`apply_parts`) they make it scan
        in the tiled chain's order."""
        return [(ph, r) for rank in range(self.n) for ph, r in self.parts(rank)]

    def touched(self, rank, phase):
        """Cells rank's part of `phase` reads and writes, or None."""
        for ph, r in self.parts(rank):
            if ph == phase:
                return self._grown(r, 1)
        return None

    def halo_entries(self, rank, phase):
        """[(peer, send rectangle or None, receive rectangle or None)] after `phase`:
        rank sends the cells it touched that the peer uses, and receives likewise."""
        out = []
        for peer in range(self.n):
            if peer == rank:
                continue
            send = _intersect(self.touched(rank, phase), self.used(peer))
            recv = _intersect(self.touched(peer, phase), self.used(rank))
            if send is not None or recv is not None:
                out.append((peer, send, recv))
        return out

    def param_entries(self, rank):
        """Parameter gather before a from-scratch residual: the owners' current
        parameters of the spaxels in rank's frame."""
        out = []
        for peer in range(self.n):
            if peer == rank:
                continue
            send = _intersect(self.owned(rank), self.region(peer))
            recv = _intersect(self.owned(peer), self.region(rank))
            if send is not None or recv is not None:
                out.append((peer, send, recv))
        return out

    def check_disjoint(self):
        """The defining property (tests): within a phase no two ranks touch a common cell."""
        for ph in self.phases:
            seen = np.zeros((self.H, self.W), dtype=np.int32)
            for rank in range(self.n):
                t = self.touched(rank, ph)
                if t is not None:
                    seen[t[0]:t[1], t[2]:t[3]] += 1
            if seen.max() > 1:
                return False
        return True


def _local(rect, origin):
    if rect is None:
        return (0, 0, 0, 0)
    return (rect[0] - origin[0], rect[1] - origin[0], rect[2] - origin[2], rect[3] - origin[2])


def plan_tables(layout, rank):
    """{plan id: int array [n, 10]} of rank's halo plans in LOCAL coordinates, the
    form d3d_halo_plan takes: (peer, kind, send y0,y1,x0,x1, recv y0,y1,x0,x1)."""
    origin = layout.region(rank)
    plans = {}
    for ph in layout.phases:
        rows = [(peer, 0) + _local(s, origin) + _local(r, origin)
                for peer, s, r in layout.halo_entries(rank, ph)]
        plans[ph] = np.array(rows, dtype=np.int32).reshape(-1, 10)
    rows = [(peer, 1) + _local(s, origin) + _local(r, origin)
            for peer, s, r in layout.param_entries(rank)]
    plans[PLAN_PARAMS] = np.array(rows, dtype=np.int32).reshape(-1, 10)
    return plans


def apply_parts(engine, layout):
    """Make a full-cube (single device) engine scan in the tiled chain's order."""
    parts = layout.all_parts()
    engine.set_parts(np.array([r for _, r in parts], dtype=np.int32),
                     np.array([ph for ph, _ in parts], dtype=np.int32))


def setup_tile_engine(engine, layout, rank):
    """Parts and halo plans of `rank` on an engine whose cube is layout.region(rank)
    (set_tile already called)."""
    origin = layout.region(rank)
    parts = layout.parts(rank)
    engine.set_parts(np.array([_local(r, origin) for _, r in parts], dtype=np.int32),
                     np.array([ph for ph, _ in parts], dtype=np.int32))
    for plan, table in plan_tables(layout, rank).items():
        engine.halo_plan(plan, table)


def make_tile_engine(layout, rank, data, var, mask, fsf, lsf, params, min_b, max_b,
                     jump_amplitude, ra, seed, device=0, err=None, refresh_every=0, options=None):
    """`_lib.Engine` holding rank's region of the global problem.  The initial
    residual is rebuilt from the region's own parameters (exact on the used cells);
    `err`, a GLOBAL residual cube, overrides it (the bit-identity tests hand every
    tile the very same starting residual as the single device)."""
    from . import _lib
    ry0, ry1, rx0, rx1 = layout.region(rank)
    oy0, oy1, ox0, ox1 = layout.owned(rank)
    D = data.shape[0]
    eng = _lib.Engine((D, ry1 - ry0, rx1 - rx0), fsf.shape, device=device, options=options)
    eng.set_taps(fsf, lsf)
    eng.set_tile(ry0, rx0, layout.W, oy0 - ry0, oy1 - ry0, ox0 - rx0, ox1 - rx0)
    sub = (slice(None), slice(ry0, ry1), slice(rx0, rx1))
    eng.set_data(np.ascontiguousarray(data[sub]),
                 None if var is None else np.ascontiguousarray(var[sub]),
                 mask=np.ascontiguousarray(np.asarray(mask)[ry0:ry1, rx0:rx1]))
    setup_tile_engine(eng, layout, rank)
    eng.set_params(np.ascontiguousarray(params[ry0:ry1, rx0:rx1]))
    eng.mh_config(min_b, max_b, jump_amplitude, ra, seed=seed, refresh_every=refresh_every)
    if err is not None:
        eng.upload_slot(_lib.SLOT_ERR, np.ascontiguousarray(err[sub]))
    else:
        eng.residual(fetch=False)
    return eng


# ---- transports -----------------------------------------------------------------

def _peer_entry(tables, peer, plan, me):
    """Index of `peer`'s entry that talks to `me` in `plan`."""
    t = tables[peer][plan]
    idx = np.nonzero(t[:, 0] == me)[0]
    return int(idx[0]) if len(idx) else None


def exchange_loopback(engines, tables, plan, device_copy=None):
    """All tiles in one process: pack everywhere, copy send buffers into the peers'
    receive buffers (device to device when `device_copy`, else through the host),
    unpack everywhere."""
    for eng in engines:
        eng.halo_pack(plan)
    for eng in engines:
        eng.sync()          # nobody still reads the receive buffers of the previous exchange
    for me, eng in enumerate(engines):
        for k, row in enumerate(tables[me][plan]):
            peer = int(row[0])
            if row[3] <= row[2] or row[5] <= row[4]:
                continue                                   # nothing sent to this peer
            j = _peer_entry(tables, peer, plan, me)
            if device_copy:
                sp, sb, _, _ = eng.halo_buffers(plan, k)
                _, _, rp, rb = engines[peer].halo_buffers(plan, j)
                assert sb == rb, "halo plans of ranks %d and %d disagree" % (me, peer)
                eng.device_copy(rp, sp, sb)
            else:
                engines[peer].halo_upload(plan, j, eng.halo_download(plan, k))
    for eng in engines:
        eng.halo_unpack(plan)


def sweep_loopback(engines, layout, tables, sweep, device_copy=None, refresh=False):
    """One sweep of every tile (one GPU, or the CPU tests): phase by phase, the halo
    copies in between; `refresh` rebuilds the residual afterwards (lib/run.py:521-534)."""
    for ph in layout.phases:
        for eng in engines:
            eng.mh_phase(ph, sweep)
        exchange_loopback(engines, tables, ph, device_copy)
    if refresh:
        exchange_loopback(engines, tables, PLAN_PARAMS, device_copy)
        for eng in engines:
            eng.residual(fetch=False)


def exchange_distributed(engine, table, plan, dist, torch):
    """One rank of a torch.distributed job, host-staged (gloo rehearsal): the packed
    send buffers go through host tensors."""
    engine.halo_pack(plan)
    ops, inbox, keep = [], [], []
    for k, row in enumerate(table):
        peer = int(row[0])
        n_recv = max(row[7] - row[6], 0) * max(row[9] - row[8], 0)
        if n_recv:
            _, _, _, rb = engine.halo_buffers(plan, k)
            buf = torch.empty(rb // 8, dtype=torch.float64)
            inbox.append((k, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        if row[3] > row[2] and row[5] > row[4]:
            t = torch.from_numpy(engine.halo_download(plan, k))
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for k, buf in inbox:
        engine.halo_upload(plan, k, buf.numpy())
    engine.halo_unpack(plan)


def sweep_distributed(engine, layout, tables, sweep, dist, torch, refresh=False, timed=False):
    """One sweep with host-staged halos (gloo rehearsal).  timed: returns the seconds the
    exchanges took on this rank (the phase's kernels drained first)."""
    import time
    spent = 0.0
    for ph in layout.phases:
        engine.mh_phase(ph, sweep)
        if timed:
            engine.sync()
            t0 = time.perf_counter()
        exchange_distributed(engine, tables[ph], ph, dist, torch)
        if timed:
            engine.sync()
            spent += time.perf_counter() - t0
    if refresh:
        exchange_distributed(engine, tables[PLAN_PARAMS], PLAN_PARAMS, dist, torch)
        engine.residual(fetch=False)
    return spent


def gather_params(layout, rank, engine):
    """(global rectangle, parameters) of the spaxels rank owns."""
    ry0, ry1, rx0, rx1 = layout.region(rank)
    oy0, oy1, ox0, ox1 = layout.owned(rank)
    p = engine.get_params()
    return (oy0, oy1, ox0, ox1), p[oy0 - ry0:oy1 - ry0, ox0 - rx0:ox1 - rx0]


def tile_grid_for(world, H=None, fh=None):
    """Row strips (world x 1): two phases per sweep and contiguous halos.  2-D grids
    (2x2, 2x4) are available through `--tiles` / TileLayout(ty, tx)."""
    return world, 1


def parse_tiles(text, world):
    if not text:
        return tile_grid_for(world)
    ty, tx = [int(v) for v in text.lower().split("x")]
    if ty * tx != world:
        raise ValueError("--tiles %s does not match %d ranks" % (text, world))
    return ty, tx


def rank_classes(layout):
    """Ranks grouped by what they compute: {((phase, rows, columns), ...): [ranks]}.  Ranks of
    one class run the same launches (up to where the cube's edge clips their windows)."""
    out = {}
    for rank in range(layout.n):
        key = tuple((ph, r[1] - r[0], r[3] - r[2]) for ph, r in layout.parts(rank))
        out.setdefault(key, []).append(rank)
    return out


def time_phases(engine, layout, rank, first_sweep, n):
    """{phase: ms per sweep} of rank's own parts, phase by phase, alone on its GPU and without
    halo traffic (a device synchronisation after every phase; the chain state moves on: for
    timing only, after the chain has been used)."""
    import time
    mine = sorted({ph for ph, _ in layout.parts(rank)})
    out = {ph: 0.0 for ph in mine}
    engine.sync()
    for s in range(first_sweep, first_sweep + n):
        for ph in mine:
            t0 = time.perf_counter()
            engine.mh_phase(ph, s)
            engine.sync()
            out[ph] += (time.perf_counter() - t0) * 1e3 / n
    return out


def critical_path(phase_ms_by_rank):
    """A tiled sweep's compute time: the phases run one after the other, all ranks at once, so
    the sweep takes  sum over phases of the SLOWEST rank's time in that phase  (a rank without
    a part in a phase waits).  phase_ms_by_rank: {rank: {phase: ms}}.
    Returns (ms, {phase: (slowest rank, ms)}, rank with the largest own total)."""
    phases = sorted({ph for t in phase_ms_by_rank.values() for ph in t})
    per_phase = {}
    for ph in phases:
        rank = max(phase_ms_by_rank, key=lambda r: phase_ms_by_rank[r].get(ph, 0.0))
        per_phase[ph] = (rank, phase_ms_by_rank[rank].get(ph, 0.0))
    total = sum(v[1] for v in per_phase.values())
    busiest = max(phase_ms_by_rank, key=lambda r: sum(phase_ms_by_rank[r].values()))
    return total, per_phase, busiest


def reference_single_context(layout, data, var, mask, fsf, lsf, init, min_b, max_b, ra, seed,
                             sweeps, device=0, refresh_every=1000):
    """The same chain on ONE context given the tiling's parts (apply_parts): what every
    tiled run must reproduce bit for bit.  Returns (parameters, accepted)."""
    from . import _lib
    D = data.shape[0]
    with _lib.Engine((D, layout.H, layout.W), fsf.shape, device=device) as ref:
        ref.set_taps(fsf, lsf)
        ref.set_data(data, var, mask=mask)
        apply_parts(ref, layout)
        ref.set_params(init)
        ref.mh_config(min_b, max_b, 0.1, ra, seed=seed, refresh_every=refresh_every)
        ref.residual(fetch=False)
        accepted = ref.mh_sweeps(sweeps, 1)
        return ref.get_params(), accepted


def bench_tiled(args, rank, local_rank, world, dist, torch):
    """`bench.py --mode tiled`: one 300x300x128 chain cut over the ranks.  With the
    nccl backend the halos travel by RCCL inside the library (d3d_mh_sweeps does
    whole sweeps on the device); with gloo they are staged through the host.

    The run verifies itself: after the timed sweeps the owners' parameters are gathered
    on rank 0 and compared with ONE context given the same parts and the same sweeps
    there -- `bit_identical` in the record; `halo_ms_per_sweep` times the exchanges
    (HIP events around pack / send-recv / unpack inside the library), `tiles` is the grid
    and `rccl_ranks` what ncclCommCount reports."""
    import json
    import time

    from . import _lib
    import bench as B

    D, H, W, fs = B.WORKLOADS[args.workload]
    fsf, lsf = B.build_taps(D, fs)
    ty, tx = parse_tiles(getattr(args, "tiles", None), world)
    layout = TileLayout(H, W, fsf.shape[0], fsf.shape[1], ty, tx)
    # every rank builds the same global synthetic problem (seed 12345)
    mask = np.ones((H, W))
    with _lib.Engine((D, H, W), fsf.shape, device=local_rank) as full:
        full.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
    ra = float(max_b[0] ** 2)
    eng = make_tile_engine(layout, rank, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                           12345, device=local_rank, refresh_every=1000,
                           options={"halo_timing": 1})
    if rank != 0:
        del data, var
    tables = plan_tables(layout, rank)
    rccl = args.backend == "nccl"
    rccl_ranks = None
    if rccl:
        box = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.comm_init(world, rank, box[0])
        rccl_ranks, rccl_rank = eng.comm_info()
        assert rccl_rank == rank
    device = torch.device("cuda", local_rank) if rccl else None
    halo_host_s = [0.0]

    def barrier():
        eng.sync()
        dist.barrier()
        eng.sync()

    def run(first, n):
        if rccl:
            return eng.mh_sweeps(n, first)
        eng.mh_accepted(reset=True)
        for s in range(first, first + n):
            halo_host_s[0] += sweep_distributed(eng, layout, tables, s, dist, torch,
                                                refresh=(s % 1000 == 0), timed=True)
        return eng.mh_accepted()

    sweep = 1
    accepted_total = 0
    if args.warmup > 0:
        accepted_total += run(sweep, args.warmup)
        sweep += args.warmup
    eng.halo_time(reset=True)
    halo_host_s[0] = 0.0
    barrier()
    t0 = time.perf_counter()
    accepted = run(sweep, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    accepted_total += accepted
    halo_ms, halo_n = eng.halo_time()
    if not rccl:
        halo_ms, halo_n = halo_host_s[0] * 1e3, args.steps * len(layout.phases)
    t = torch.tensor([dt, halo_ms], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, halo_ms = float(t[0].item()), float(t[1].item())
    a = torch.tensor([float(accepted), float(accepted_total)], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    halo_bytes = sum(int((r[3] - r[2]) * (r[5] - r[4])) * eng.shape[0] * 8
                     for ph in layout.phases for r in tables[ph])
    # ---- self-verification: the owners' parameters against one context with the same parts
    mine = gather_params(layout, rank, eng)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    verdict = None
    if rank == 0:
        got = np.full((H, W, 3), np.nan)
        for (y0, y1, x0, x1), p in everyone:
            got[y0:y1, x0:x1] = p
        n_sweeps = args.warmup + args.steps
        want, want_acc = reference_single_context(layout, data, var, mask, fsf, lsf, init, min_b,
                                                  max_b, ra, 12345, n_sweeps, device=local_rank)
        verdict = bool(np.array_equal(got, want)) and int(a[1].item()) == int(want_acc)
        max_diff = float(np.nanmax(np.abs(got - want)))
        if not verdict:
            bad = int(np.sum(np.any(got != want, axis=2)))
            sys_err = "tiled chain differs from the single context: %d spaxels, accepted %d vs %d" % (
                bad, int(a[1].item()), int(want_acc))
            import sys
            sys.stderr.write("bench.py --mode tiled: %s\n" % sys_err)
    # ---- the compute side of the sweep, rank by rank (the chain is not used after this) ----
    # every rank times its own phases alone; the sweep's critical path is the sum over the
    # phases of the slowest rank (critical_path): what the measured ms_per_step is made of
    # beside the halo copies
    dist.barrier()
    my_phases = time_phases(eng, layout, rank, sweep + args.steps, 3)
    all_phases = [None] * world
    dist.all_gather_object(all_phases, my_phases)
    cp_ms, cp_by_phase, busiest = critical_path(dict(enumerate(all_phases)))
    out = {
        "metric": "spaxel-updates/sec (MH-Gibbs)", "value": round(args.steps * H * W / dt, 1),
        "unit": "spaxel-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "cube": [D, H, W],
                   "parallelism": "1 chain tiled %dx%d, %d phases per sweep, halo copies over %s"
                                  % (ty, tx, len(layout.phases),
                                     "RCCL send/recv (in-library, device buffers)" if rccl
                                     else "gloo (host-staged rehearsal)"),
                   "halo_bytes_sent_per_sweep_rank0": halo_bytes},
        "acceptance": round(float(a[0].item()) / float(args.steps * H * W), 4),
        "tiles": [ty, tx],
        "phases_per_sweep": len(layout.phases),
        "rccl_ranks": rccl_ranks,
        "halo_ms_per_sweep": round(halo_ms / max(args.steps, 1), 4),
        "halo_exchanges_timed": int(halo_n),
        "bit_identical": verdict,
        # (where the tiles' own from-scratch residuals are not bit-identical to the full
        # cube's -- depths whose convolution kernel sums in a position-dependent order,
        # DESIGN.md section 7 -- the chains agree to rounding and this is their distance)
        "max_abs_param_diff": max_diff if rank == 0 else None,
        "verified_against": "one context given the same parts (tiling.apply_parts), %d sweeps, on rank 0"
                            % (args.warmup + args.steps),
        # sum over the phases of the slowest rank's own compute time (no halo copies): the
        # floor of ms_per_step on this grid, and the rank whose parts make it
        "projected_critical_path_ms": round(cp_ms, 4),
        "slowest_rank": int(busiest),
        "critical_path_by_phase": {str(ph): {"rank": int(r), "ms": round(ms, 4)}
                                   for ph, (r, ms) in cp_by_phase.items()},
    }
    if rccl:
        eng.comm_destroy()
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def bench_tiled_loopback(args, device):
    """`bench.py --mode tiled --gpus 1 --tiles TYxTX`: the tiled chain with every tile
    a context of THIS process on one GPU, halos copied device to device -- the
    rehearsal of config 4 a one-GPU box allows (the tiles share the GPU, so this
    measures the protocol's overhead, not a speed-up)."""
    import json
    import time

    from . import _lib
    import bench as B

    D, H, W, fs = B.WORKLOADS[args.workload]
    fsf, lsf = B.build_taps(D, fs)
    ty, tx = [int(v) for v in (args.tiles or "2x1").lower().split("x")]
    layout = TileLayout(H, W, fsf.shape[0], fsf.shape[1], ty, tx)
    mask = np.ones((H, W))
    with _lib.Engine((D, H, W), fsf.shape, device=device) as full:
        full.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
    ra = float(max_b[0] ** 2)
    engines = [make_tile_engine(layout, r, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                                12345, device=device) for r in range(layout.n)]
    tables = [plan_tables(layout, r) for r in range(layout.n)]
    sweep = 1
    for _ in range(args.warmup):
        sweep_loopback(engines, layout, tables, sweep, device_copy=True)
        sweep += 1
    for e in engines:
        e.sync()
        e.mh_accepted(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep_loopback(engines, layout, tables, sweep, device_copy=True)
        sweep += 1
    for e in engines:
        e.sync()
    dt = time.perf_counter() - t0
    accepted = sum(e.mh_accepted() for e in engines)
    # self-verification, as bench_tiled: against one context given the same parts
    got = np.full((H, W, 3), np.nan)
    for r in range(layout.n):
        (y0, y1, x0, x1), p = gather_params(layout, r, engines[r])
        got[y0:y1, x0:x1] = p
    want, _ = reference_single_context(layout, data, var, mask, fsf, lsf, init, min_b, max_b, ra,
                                       12345, args.warmup + args.steps, device=device,
                                       refresh_every=0)
    # what each tile would compute per sweep alone on a GPU, phase by phase, and the critical
    # path of the grid (one representative of every class of ranks is timed)
    timed = {}
    for ranks in rank_classes(layout).values():
        t = time_phases(engines[ranks[0]], layout, ranks[0], sweep, 3)
        for r in ranks:
            timed[r] = t
    cp_ms, cp_by_phase, busiest = critical_path(timed)
    out = {
        "metric": "spaxel-updates/sec (MH-Gibbs)", "value": round(args.steps * H * W / dt, 1),
        "unit": "spaxel-updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "tiles": [ty, tx], "phases_per_sweep": len(layout.phases), "rccl_ranks": None,
        "projected_critical_path_ms": round(cp_ms, 4), "slowest_rank": int(busiest),
        "critical_path_by_phase": {str(ph): {"rank": int(r), "ms": round(ms, 4)}
                                   for ph, (r, ms) in cp_by_phase.items()},
        "bit_identical": bool(np.array_equal(got, want)),
        "max_abs_param_diff": float(np.nanmax(np.abs(got - want))),
        "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "cube": [D, H, W],
                   "parallelism": "1 chain tiled %dx%d as %d contexts of one process on ONE GPU "
                                  "(loop-back rehearsal), %d phases per sweep, halos copied device "
                                  "to device" % (ty, tx, layout.n, len(layout.phases))},
        "acceptance": round(accepted / float(args.steps * H * W), 4),
    }
    for e in engines:
        e.close()
    print(json.dumps(out))

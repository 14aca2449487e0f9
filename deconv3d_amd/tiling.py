# coding=utf-8
"""
Spatial tiling of ONE chain over several GPUs (BASELINE config 4).

The reference is a single process; nothing here has a counterpart in it.  An
update at spaxel (y, x) touches only its FSF window (lib/run.py:404-419), so the
cube is cut into ty x tx tiles of owned spaxels.  A tile's device context holds
its owned rectangle plus a frame of FSF half-width cells.  Colour classes and
random numbers are keyed by GLOBAL coordinates, hence every rank makes exactly
the decisions the single-device chain makes; what a rank cannot compute itself
-- the residual change caused by a NEIGHBOUR's update whose window reaches into
its region -- is replayed from 8-double records {y, x, a,c,w before, a,c,w after}
(`d3d_export_updates` / `d3d_apply_updates`).  The tiled chain is therefore
bit-identical to the single-device chain (tests/test_gpu_tiling.py,
tests/test_tiling_cpu.py).

Exchange pattern: after each colour class every rank sends the records of its
owned spaxels lying within two FSF half-widths of a neighbour's owned rectangle
(<= 8 neighbours, point-to-point, a few hundred bytes) -- over RCCL these ride a
direct xGMI link each; there is no ring collective on the path.  The price is
fh*fw dependent exchanges per sweep: at 300x300 the tiled chain is
latency-bound and slower than one GPU; it exists for cubes/chains that must be
split, not as the throughput mode (that is the ensemble, bench.py).

Engines are duck-typed: `mh_colour(colour, sweep)`, `export_updates(idx)`,
`apply_updates(records)`.  The product engine is `_lib.Engine`; the CPU tests
drive the same code with an oracle-backed engine.
"""
from __future__ import annotations

import numpy as np


def _splits(n, parts):
    """Boundaries of `parts` nearly equal consecutive ranges covering [0, n)."""
    base, extra = divmod(n, parts)
    edges = [0]
    for i in range(parts):
        edges.append(edges[-1] + base + (1 if i < extra else 0))
    return edges


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

    def owned(self, rank):
        """Global (y0, y1, x0, x1) of the spaxels rank owns."""
        iy, ix = divmod(rank, self.tx)
        return (self.row_edges[iy], self.row_edges[iy + 1],
                self.col_edges[ix], self.col_edges[ix + 1])

    def region(self, rank):
        """Global (y0, y1, x0, x1) of the cells rank stores: owned + FSF half widths."""
        y0, y1, x0, x1 = self.owned(rank)
        return (max(y0 - self.fhh, 0), min(y1 + self.fhh, self.H),
                max(x0 - self.fhw, 0), min(x1 + self.fhw, self.W))

    def _near(self, rank, other):
        """Boolean (H, W) map of rank's owned spaxels whose window reaches into
        other's region, i.e. within 2 half-widths of other's owned rectangle."""
        y0, y1, x0, x1 = self.owned(rank)
        oy0, oy1, ox0, ox1 = self.owned(other)
        ys = np.arange(self.H)
        xs = np.arange(self.W)
        ymask = (ys >= y0) & (ys < y1) & (ys >= oy0 - 2 * self.fhh) & (ys < oy1 + 2 * self.fhh)
        xmask = (xs >= x0) & (xs < x1) & (xs >= ox0 - 2 * self.fhw) & (xs < ox1 + 2 * self.fhw)
        return ymask[:, None] & xmask[None, :]

    def neighbours(self, rank):
        return [o for o in range(self.n) if o != rank and self._near(rank, o).any()]

    def send_lists(self, rank, mask):
        """{neighbour: [global (n_c, 2) int arrays of (y, x), one per colour]} of the
        unmasked spaxels rank must report, in the order the colour is scanned
        (row-major inside the colour)."""
        out = {}
        live = np.asarray(mask) == 1
        for nb in self.neighbours(rank):
            sel = self._near(rank, nb) & live
            per_colour = []
            for cy in range(self.fh):
                for cx in range(self.fw):
                    ys, xs = np.nonzero(sel[cy::self.fh, cx::self.fw])
                    per_colour.append(np.stack((cy + ys * self.fh, cx + xs * self.fw), axis=1)
                                      .astype(np.int64))
            out[nb] = per_colour
        return out


def make_tile_engine(layout, rank, data, var, mask, fsf, lsf, params, min_b, max_b,
                     jump_amplitude, ra, seed, device=0, err=None):
    """`_lib.Engine` holding rank's region of the global problem.  `err` is the
    GLOBAL initial residual (cells near a region border receive contributions
    from spaxels outside the region, so it cannot be rebuilt from the region's
    own parameters); when None it is computed with a temporary full-size
    context on the same device."""
    from . import _lib
    if err is None:
        with _lib.Engine(data.shape, fsf.shape, device=device) as full:
            full.set_taps(fsf, lsf)
            full.set_data(data, var, mask=mask)
            full.set_params(params)
            err = full.residual()
    ry0, ry1, rx0, rx1 = layout.region(rank)
    oy0, oy1, ox0, ox1 = layout.owned(rank)
    D = data.shape[0]
    eng = _lib.Engine((D, ry1 - ry0, rx1 - rx0), fsf.shape, device=device)
    eng.set_taps(fsf, lsf)
    eng.set_tile(ry0, rx0, layout.W, oy0 - ry0, oy1 - ry0, ox0 - rx0, ox1 - rx0)
    sub = (slice(None), slice(ry0, ry1), slice(rx0, rx1))
    eng.set_data(np.ascontiguousarray(data[sub]), np.ascontiguousarray(var[sub]),
                 mask=np.ascontiguousarray(np.asarray(mask)[ry0:ry1, rx0:rx1]))
    eng.set_params(np.ascontiguousarray(params[ry0:ry1, rx0:rx1]))
    eng.mh_config(min_b, max_b, jump_amplitude, ra, seed=seed, refresh_every=0)
    eng.upload_slot(_lib.SLOT_ERR, np.ascontiguousarray(err[sub]))
    return eng


class TileStepper(object):
    """One rank's side of the per-colour protocol."""

    def __init__(self, layout, rank, engine, mask):
        self.layout, self.rank, self.engine = layout, rank, engine
        self.region = layout.region(rank)
        ry0, ry1, rx0, rx1 = self.region
        self.local_w = rx1 - rx0
        self.send = layout.send_lists(rank, mask)            # what I report, per neighbour
        self.recv_counts = {nb: [len(a) for a in layout.send_lists(nb, mask)[rank]]
                            for nb in layout.neighbours(rank)
                            if rank in layout.neighbours(nb)}

    def update(self, colour, sweep):
        """Update my spaxels of `colour`; returns {neighbour: records[n, 8]} with
        GLOBAL coordinates in the first two columns."""
        self.engine.mh_colour(colour, sweep)
        ry0, _, rx0, _ = self.region
        out = {}
        for nb, per_colour in self.send.items():
            yx = per_colour[colour]
            if len(yx) == 0:
                continue
            idx = (yx[:, 0] - ry0) * self.local_w + (yx[:, 1] - rx0)
            out[nb] = self.engine.export_updates(idx.astype(np.int32))
        return out

    def replay(self, records):
        """Apply a neighbour's records (global coordinates) to my region."""
        if records is None or len(records) == 0:
            return
        rec = np.array(records, dtype=np.float64).reshape(-1, 8)
        rec[:, 0] -= self.region[0]
        rec[:, 1] -= self.region[2]
        self.engine.apply_updates(rec)


def sweep_loopback(steppers, sweep, ncolours):
    """All tiles in one process (one GPU, or the CPU tests): per colour every
    tile updates, then every tile replays what the others reported."""
    for colour in range(ncolours):
        outbox = [st.update(colour, sweep) for st in steppers]
        for src, msgs in enumerate(outbox):
            for dst, rec in msgs.items():
                steppers[dst].replay(rec)


def sweep_distributed(stepper, sweep, ncolours, dist, torch, device=None):
    """One rank of a torch.distributed job (backend nccl = RCCL over xGMI, or
    gloo): point-to-point exchange of the border records after each colour."""
    rank = stepper.rank
    for colour in range(ncolours):
        msgs = stepper.update(colour, sweep)
        ops, inbox = [], []
        for nb, counts in stepper.recv_counts.items():
            n = counts[colour]
            if n:
                buf = torch.empty((n, 8), dtype=torch.float64, device=device)
                inbox.append(buf)
                ops.append(dist.P2POp(dist.irecv, buf, nb))
        keep = []
        for nb, rec in msgs.items():
            t = torch.from_numpy(np.ascontiguousarray(rec))
            if device is not None:
                t = t.to(device)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, nb))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for buf in inbox:
            stepper.replay(buf.cpu().numpy())
    del rank


def gather_params(layout, rank, engine):
    """(global rectangle, parameters) of the spaxels rank owns."""
    ry0, ry1, rx0, rx1 = layout.region(rank)
    oy0, oy1, ox0, ox1 = layout.owned(rank)
    p = engine.get_params()
    return (oy0, oy1, ox0, ox1), p[oy0 - ry0:oy1 - ry0, ox0 - rx0:ox1 - rx0]


def tile_grid_for(world):
    """ty x tx for `world` ranks: 2 -> 1x2, 4 -> 2x2, 8 -> 2x4 (config 4)."""
    ty = int(np.floor(np.sqrt(world)))
    while world % ty:
        ty -= 1
    return ty, world // ty


def bench_tiled(args, rank, local_rank, world, dist, torch):
    """`bench.py --mode tiled`: one 300x300x128 chain cut over the ranks."""
    import json
    import time

    from . import _lib
    import bench as B

    D, H, W, fs = B.WORKLOADS[args.workload]
    fsf, lsf = B.build_taps(D, fs)
    ty, tx = tile_grid_for(world)
    layout = TileLayout(H, W, fsf.shape[0], fsf.shape[1], ty, tx)
    # every rank builds the same global synthetic problem (seed 12345)
    mask = np.ones((H, W))
    with _lib.Engine((D, H, W), fsf.shape, device=local_rank) as full:
        full.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
        full.set_data(data, var, mask=mask)
        full.set_params(init)
        err0 = full.residual()
    ra = float(max_b[0] ** 2)
    eng = make_tile_engine(layout, rank, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                           12345, device=local_rank, err=err0)
    stepper = TileStepper(layout, rank, eng, mask)
    ncol = fsf.shape[0] * fsf.shape[1]
    device = torch.device("cuda", local_rank) if args.backend == "nccl" else None

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    sweep = 1
    for _ in range(args.warmup):
        sweep_distributed(stepper, sweep, ncol, dist, torch, device)
        sweep += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep_distributed(stepper, sweep, ncol, dist, torch, device)
        sweep += 1
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    out = {
        "metric": "spaxel-updates/sec (MH-Gibbs)", "value": round(args.steps * H * W / dt, 1),
        "unit": "spaxel-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "cube": [D, H, W],
                   "parallelism": "1 chain tiled %dx%d, border-update replay over %s"
                                  % (ty, tx, args.backend)},
    }
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))

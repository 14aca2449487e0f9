"""Oracle-backed tile engine: the duck type deconv3d_amd.tiling drives (mh_phase,
halo_pack / halo_unpack / halo_download / halo_upload / halo_buffers, residual),
on the CPU.  Test infrastructure."""
import numpy as np

from deconv3d_amd import tiling
from oracle import deconv3d_oracle as O


def part_order(parts, phase, mask, fh, fw, origin=(0, 0)):
    """(y, x) (local) of the unmasked spaxels of the parts of `phase`, in the device's
    scan order: part by part, GLOBAL colour class by colour class, row-major inside."""
    gy0, gx0 = origin
    for ph, (y0, y1, x0, x1) in parts:
        if ph != phase:
            continue
        for cy in range(fh):
            for cx in range(fw):
                for y in range(y0, y1):
                    if (y + gy0) % fh != cy:
                        continue
                    for x in range(x0, x1):
                        if (x + gx0) % fw == cx and mask[y, x] == 1:
                            yield (y, x)


def sweep_in_part_order(st, layout, sweep):
    """The single-domain chain in the tiled chain's scan order (tiling.apply_parts)."""
    fh, fw = st.fsf.shape
    parts = layout.all_parts()
    for ph in layout.phases:
        for (y, x) in part_order(parts, ph, st.mask, fh, fw):
            O.mh_update(st, y, x, sweep)


class OracleTileEngine(object):
    def __init__(self, layout, rank, data, var, mask, fsf, lsf, params, min_b, max_b,
                 jump_amplitude, ra, seed, global_err=None):
        self.layout, self.rank = layout, rank
        ry0, ry1, rx0, rx1 = layout.region(rank)
        self.region = (ry0, ry1, rx0, rx1)
        sub = (slice(None), slice(ry0, ry1), slice(rx0, rx1))
        self.st = O.MHState(data[sub], var[sub], np.asarray(mask)[ry0:ry1, rx0:rx1], fsf, lsf,
                            params[ry0:ry1, rx0:rx1], min_b, max_b, jump_amplitude, ra, seed,
                            origin=(ry0, rx0, layout.W),
                            err=None if global_err is None else global_err[sub])
        self.parts = [(ph, (r[0] - ry0, r[1] - ry0, r[2] - rx0, r[3] - rx0))
                      for ph, r in layout.parts(rank)]
        self.tables = tiling.plan_tables(layout, rank)
        self.sendbuf, self.recvbuf = {}, {}

    def mh_phase(self, phase, sweep):
        fh, fw = self.layout.fh, self.layout.fw
        for (y, x) in part_order(self.parts, phase, self.st.mask, fh, fw, self.region[::2]):
            O.mh_update(self.st, y, x, sweep)

    # -- halos ---------------------------------------------------------------------
    def _cells(self, kind, rect):
        y0, y1, x0, x1 = [int(v) for v in rect]
        if kind == 0:     # residual cells, packed (y, x, z) like the device's layout
            return np.ascontiguousarray(np.transpose(self.st.err[:, y0:y1, x0:x1], (1, 2, 0)))
        return self.st.params[y0:y1, x0:x1].copy()

    def halo_pack(self, plan):
        for k, row in enumerate(self.tables[plan]):
            if row[3] > row[2] and row[5] > row[4]:
                self.sendbuf[(plan, k)] = self._cells(row[1], row[2:6]).ravel()

    def halo_download(self, plan, k):
        return self.sendbuf[(plan, k)]

    def halo_upload(self, plan, k, values):
        self.recvbuf[(plan, k)] = np.array(values, dtype=np.float64)

    def halo_buffers(self, plan, k):
        row = self.tables[plan][k]
        e = self.st.data.shape[0] if row[1] == 0 else 3
        ns = max(row[3] - row[2], 0) * max(row[5] - row[4], 0) * e
        nr = max(row[7] - row[6], 0) * max(row[9] - row[8], 0) * e
        return None, int(ns) * 8, None, int(nr) * 8

    def halo_unpack(self, plan):
        for k, row in enumerate(self.tables[plan]):
            y0, y1, x0, x1 = [int(v) for v in row[6:10]]
            if y1 <= y0 or x1 <= x0:
                continue
            v = self.recvbuf.pop((plan, k))
            if row[1] == 0:
                self.st.err[:, y0:y1, x0:x1] = np.transpose(
                    v.reshape(y1 - y0, x1 - x0, -1), (2, 0, 1))
            else:
                self.st.params[y0:y1, x0:x1] = v.reshape(y1 - y0, x1 - x0, 3)

    def residual(self, fetch=False):
        self.st.err = O.compute_error_in_one_step(self.st.data, self.st.params, self.st.mask,
                                                  self.st.fsf, self.st.lsf)
        return self.st.err if fetch else None

    def sync(self):
        pass

    def get_params(self):
        return self.st.params

    def mh_accepted(self, reset=False):
        n = self.st.accepted
        if reset:
            self.st.accepted = 0
        return n

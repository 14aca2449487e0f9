"""Oracle-backed tile engine: the duck type deconv3d_amd.tiling drives
(mh_colour / export_updates / apply_updates), on the CPU.  Test infrastructure."""
import numpy as np

from oracle import deconv3d_oracle as O


class OracleTileEngine(object):
    def __init__(self, layout, rank, data, var, mask, fsf, lsf, params, min_b, max_b,
                 jump_amplitude, ra, seed, global_err):
        self.layout = layout
        ry0, ry1, rx0, rx1 = layout.region(rank)
        self.region = (ry0, ry1, rx0, rx1)
        self.owned = layout.owned(rank)
        sub = (slice(None), slice(ry0, ry1), slice(rx0, rx1))
        self.st = O.MHState(data[sub], var[sub], np.asarray(mask)[ry0:ry1, rx0:rx1], fsf, lsf,
                            params[ry0:ry1, rx0:rx1], min_b, max_b, jump_amplitude, ra, seed,
                            origin=(ry0, rx0, layout.W), err=global_err[sub])
        self.prev = np.zeros_like(self.st.params)

    def mh_colour(self, colour, sweep):
        fh, fw = self.layout.fh, self.layout.fw
        cy, cx = divmod(colour, fw)
        ry0, ry1, rx0, rx1 = self.region
        oy0, oy1, ox0, ox1 = self.owned
        for gy in range(oy0, oy1):
            if gy % fh != cy:
                continue
            for gx in range(ox0, ox1):
                if gx % fw != cx or self.st.mask[gy - ry0, gx - rx0] != 1:
                    continue
                O.mh_update(self.st, gy - ry0, gx - rx0, sweep)
                self.prev[gy - ry0, gx - rx0] = self.st.last[0]

    def export_updates(self, idx):
        W = self.region[3] - self.region[2]
        out = np.empty((len(idx), 8))
        for i, sp in enumerate(idx):
            y, x = divmod(int(sp), W)
            out[i, 0] = y + self.region[0]
            out[i, 1] = x + self.region[2]
            out[i, 2:5] = self.prev[y, x]
            out[i, 5:8] = self.st.params[y, x]
        return out

    def apply_updates(self, records):
        for rec in np.asarray(records).reshape(-1, 8):
            O.replay_update(self.st, int(rec[0]), int(rec[1]), rec[2:5], rec[5:8])

    def get_params(self):
        return self.st.params

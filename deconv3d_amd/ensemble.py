"""
Several independent chains on ONE GPU (the one-device form of BASELINE config 5, the
ensemble: lib/run.py has one chain per Run()).

A colour launch of a small cube leaves most of the chip idle (config 2: 49 workgroups for
1024 slots), and its 12 us are latency, not bandwidth.  Independent contexts run on their
own HIP streams, so their launches overlap when they are ENQUEUED concurrently: one host
thread per chain (ctypes releases the GIL during a call).  Measured on MI355X
(profiles/r03_replicas.txt): config 2, 64x64x64 -- 1 / 2 / 4 chains 2.7 / 5.2 / 9.5 M
spaxel-updates/s in total (the runtime's four hardware queues bound it: 8 chains 9.5 M);
config 1 -- 0.35 / 0.68 / 1.32 M.  A chip-filling cube gains nothing (round 1: 14.9 vs
15.3 M for two 300x300x128 chains).  Every chain is exactly the chain its context would
produce alone: contexts share nothing.
"""
import threading

from . import _lib


def sweep_chains_batched(engines, n_sweeps, first_sweep=1, keep_one_in=1, chains=None, dlogs=None):
    """The chains of several engines of ONE geometry (same shape, mask, FSF, LSF; data,
    variance values, bounds, start and seed may differ) in one launch per colour class
    (d3d_mh_sweeps_batch): a small cube's launch carries len(engines) times the windows
    for the same latency, beyond what concurrent streams give (config 2: see
    profiles/r03_replicas.txt).  Cubes up to 256 channels, unpartitioned.  Returns the
    accepted counts; chains / dlogs: per-engine host arrays (or None) that receive the saved
    sweeps exactly as Engine.mh_sweeps fills them."""
    return _lib.mh_sweeps_batch(list(engines), n_sweeps, first_sweep, keep_one_in, chains, dlogs)


def sweep_chains(engines, n_sweeps, first_sweep=1, keep_one_in=1, chains=None, dlogs=None):
    """engine.mh_sweeps(n_sweeps, first_sweep, ...) of every engine, concurrently.
    chains / dlogs: per-engine host arrays as Engine.mh_sweeps takes them, or None.
    Returns the accepted counts, one per engine; re-raises the first error of a chain."""
    n = len(engines)
    out = [None] * n
    err = [None] * n

    def work(i):
        try:
            out[i] = engines[i].mh_sweeps(n_sweeps, first_sweep, keep_one_in,
                                          None if chains is None else chains[i],
                                          None if dlogs is None else dlogs[i])
        except BaseException as exc:          # noqa: BLE001 -- handed to the caller below
            err[i] = exc

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out

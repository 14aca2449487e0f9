"""Round-2 exploration: mixing of the C1 chain (calibrates the SURVEY 8(d) chain-tolerance test)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

D, H, W = 32, 16, 16
fsf = O.gaussian_fsf_image(3.0); lsf = O.gaussian_lsf_vector(D, 0.9088)
data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=12345)
ra = float(mx[0] ** 2)

def device_chain(start, seed, n, first=1):
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf); eng.set_data(data, var, mask=mask); eng.set_params(start)
        eng.mh_config(mn, mx, 0.1, ra, seed=seed, refresh_every=1000)
        ch = np.full((first + n, H, W, 3), np.nan)
        t0 = time.time(); eng.mh_sweeps(n, first, 1, ch, None); dt = time.time() - t0
        return ch[first:], dt

def bm(ch, nb):
    n = (ch.shape[0] // nb) * nb
    ch = ch[-n:]
    b = ch.reshape(nb, -1, *ch.shape[1:]).mean(1)
    return ch.mean(0), b.std(0, ddof=1) / np.sqrt(nb)

bright = truth[..., 0] > 3.0
print("bright spaxels", bright.sum())
for N in (4000, 20000):
    chains = []
    for seed in (11, 22, 33):
        ch, dt = device_chain(init, seed, N)
        chains.append(ch)
    print("N=%d sweeps from the uniform start: %.1fs per chain" % (N, dt))
    for burn_frac in (0.5,):
        b0 = int(N * burn_frac)
        stats = [bm(c[b0:], 20) for c in chains]
        for (i, j) in ((0, 1), (0, 2), (1, 2)):
            z = (stats[i][0] - stats[j][0]) / np.sqrt(stats[i][1] ** 2 + stats[j][1] ** 2)
            for sel, nm in ((bright, "bright"), (~bright, "faint")):
                print("  seeds %d/%d %s: rms z a %.2f c %.2f w %.2f | frac|z|<3 %.3f | max %.1f"
                      % (i, j, nm, *[np.sqrt(np.mean(z[..., k][sel] ** 2)) for k in range(3)],
                         np.mean(np.abs(z[sel]) < 3), np.abs(z[sel]).max()))
        # Gelman-Rubin over the three chains
        m = np.array([s[0] for s in stats]); wv = np.array([c[b0:].var(0, ddof=1) for c in chains]).mean(0)
        n = N - b0
        bvar = n * m.var(0, ddof=1)
        rhat = np.sqrt(((n - 1) / n * wv + bvar / n) / wv)
        for sel, nm in ((bright, "bright"), (~bright, "faint")):
            print("  R-hat %s: median %.3f 95%% %.3f max %.3f" % (nm, np.median(rhat[sel]),
                  np.percentile(rhat[sel], 95), rhat[sel].max()))
        post_sd = np.sqrt(wv)
        print("  posterior sd (bright) a %.3f c %.3f w %.3f ; |mean-truth|/sd median %.2f"
              % (*[np.median(post_sd[..., k][bright]) for k in range(3)],
                 np.median((np.abs(m.mean(0) - truth) / post_sd)[bright])))
    sys.stdout.flush()
# oracle from the converged state: 300 sweeps, against the long device run (different seed)
start = chains[0][-1]
t0 = time.time()
st = O.MHState(data, var, mask, fsf, lsf, start, mn, mx, seed=777)
NO = 300
och = np.empty((NO, H, W, 3))
for s in range(1, NO + 1):
    O.mh_sweep(st, s); och[s - 1] = st.params
print("oracle %d sweeps from the converged state: %.1fs" % (NO, time.time() - t0))
dev, _ = device_chain(start, 4242, 20000)
mo, so = bm(och, 10); md, sd = bm(dev, 20)
z = (mo - md) / np.sqrt(so ** 2 + sd ** 2)
for sel, nm in ((bright, "bright"), (~bright, "faint")):
    print("oracle(300) vs device(20000) %s: rms z a %.2f c %.2f w %.2f | frac|z|<3 %.3f max %.1f"
          % (nm, *[np.sqrt(np.mean(z[..., k][sel] ** 2)) for k in range(3)],
             np.mean(np.abs(z[sel]) < 3), np.abs(z[sel]).max()))

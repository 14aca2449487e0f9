"""
CPU tests (no GPU) of the boundary and the host side:
  * libdeconv3d_hip.so loads and exports every symbol include/*.h declares,
  * without a device the product fails LOUDLY (no CPU fallback),
  * the plugin classes mirror the reference's contracts (names, arguments,
    exceptions) and their taps equal the oracle's restatement of
    lib/spread_functions.py,
  * Run() input validation raises the reference's exception classes before any
    device work.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import deconv3d_amd as d3d
from deconv3d_amd import _lib
from deconv3d_amd import spread_functions as sf
from oracle import deconv3d_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INPUT = "/root/reference/tests/input"


def header_symbols():
    text = open(os.path.join(ROOT, "include", "deconv3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libdeconv3d_hip.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared, "python binding list out of sync with the header"
    lib.d3d_version.restype = ctypes.c_int
    assert lib.d3d_version() >= 100


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "deconv3d_hip.h")).read()
    assert text.count("lib/run.py:") >= 15
    assert "lib/convolution.py:89-120" in text


def _gpu():
    return _lib.device_count() > 0


@pytest.mark.skipif(_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_without_device():
    with pytest.raises(_lib.HipError, match="no CPU fallback"):
        _lib.Engine((8, 4, 4), (3, 3))
    cube = d3d.MUSE().build_cube(np.random.default_rng(0).random((8, 9, 9)))
    with pytest.raises(_lib.HipError):
        d3d.Run(cube, d3d.MUSE(), max_iterations=2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deconv3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    # ... nor do the tools; at the top level only bench.py (inside its cpu_baseline legs)
    # and __graft_entry__.smoke() may
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith((".py", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)from oracle import", bench, re.M):
        assert m.group(1), "bench.py imports the oracle at module level"
        head = bench[:m.start()].rsplit("\ndef ", 1)[1].split("(")[0]
        assert head.startswith("cpu_baseline"), head
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    for m in re.finditer(r"^(\s*)from oracle import", entry, re.M):
        head = entry[:m.start()].rsplit("\ndef ", 1)[1].split("(")[0]
        assert m.group(1) and head == "smoke", head


def test_c_abi_argument_errors_have_messages():
    lib = _lib.load()
    ctx = ctypes.c_void_p(None)
    rc = lib.d3d_ctx_create(ctypes.byref(ctx), 0, 8, 4, 4, 4, 3)     # even FSF
    assert rc == _lib.ERR_INVALID
    assert b"odd" in lib.d3d_last_error()
    rc = lib.d3d_ctx_create(ctypes.byref(ctx), 0, 0, 4, 4, 3, 3)
    assert rc == _lib.ERR_INVALID
    assert lib.d3d_sync(None) == _lib.ERR_INVALID


# ---- plugin classes --------------------------------------------------------------

def muse_cube(shape=(32, 16, 16), seed=0):
    return d3d.MUSE().build_cube(np.random.default_rng(seed).random(shape) + 1.0)


def test_instrument_type_checks():
    # lib/instruments.py:27-34
    with pytest.raises(ValueError):
        d3d.Instrument(lsf="nope", fsf=d3d.GaussianFieldSpreadFunction(1.0))
    with pytest.raises(ValueError):
        d3d.Instrument(lsf=d3d.GaussianLineSpreadFunction(1e-4), fsf=object())
    m = d3d.MUSE()
    assert isinstance(m.lsf, d3d.GaussianLineSpreadFunction) and m.lsf.fwhm == 0.0002675
    assert isinstance(m.fsf, d3d.GaussianFieldSpreadFunction) and m.fsf.fwhm == 1.0


def test_muse_cube_metadata():
    cube = muse_cube()
    assert abs(cube.get_step(1).to('arcsec').value - 0.2) < 1e-9     # lib/instruments.py:126-127
    assert abs(cube.get_step(0).to('um').value - 1.25e-4) < 1e-15
    assert cube.shape == (32, 16, 16) and not cube.is_empty()
    assert d3d.Cube().is_empty()


def test_spread_function_taps_equal_oracle_restatement():
    cube = muse_cube((32, 17, 15))
    # Gaussian FSF FWHM 0.6" at 0.2"/px = 3 px -> 9x9 (C1)
    img = d3d.GaussianFieldSpreadFunction(fwhm=0.6).as_image(cube)
    ref = O.gaussian_fsf_image(0.6 / cube.get_step(1).to('arcsec').value)
    assert img.shape == (9, 9)
    np.testing.assert_array_equal(img, ref)
    img = d3d.GaussianFieldSpreadFunction(fwhm=0.6, pa=30., ba=0.7).as_image(cube)
    np.testing.assert_array_equal(
        img, O.gaussian_fsf_image(0.6 / cube.get_step(1).to('arcsec').value, 30., 0.7))
    assert abs(img.sum() - 1) < 1e-14 and not np.allclose(img, img.T)
    # Moffat: cube-sized image as in the reference (lib/spread_functions.py:165-189)
    mof = d3d.MoffatFieldSpreadFunction(fwhm=0.6, beta=2.5).as_image(cube)
    assert mof.shape == (17, 15)
    np.testing.assert_array_equal(
        mof, O.moffat_fsf_image((17, 15), 2.5, fwhm_px=0.6 / cube.get_step(1).to('arcsec').value))
    mof11 = d3d.MoffatFieldSpreadFunction(fwhm=0.6, beta=2.5, size=11).as_image(cube)
    np.testing.assert_allclose(mof11, O.moffat_cropped(11, 3.0, 2.5), rtol=1e-12)
    alpha = d3d.MoffatFieldSpreadFunction(alpha=0.5, beta=3.0).as_image(cube)
    np.testing.assert_array_equal(alpha, O.moffat_fsf_image((17, 15), 3.0, alpha_px=0.5 / 0.2 * (0.2 / cube.get_step(1).to('arcsec').value)))
    assert d3d.NoFieldSpreadFunction().as_image(cube).shape == (17, 15)
    im = np.ones((3, 3)) / 9
    assert d3d.ImageFieldSpreadFunction(im).as_image(cube) is im
    # LSF: FWHM 2.675 A at 1.25 A/px
    lsf = d3d.GaussianLineSpreadFunction(0.0002675).as_vector(cube)
    sigma = 0.0002675 / 2.35482 / cube.get_step(0).to('um').value
    np.testing.assert_array_equal(lsf, O.gaussian_lsf_vector(32, sigma))
    assert abs(sigma - 0.9088) < 1e-3 and abs(lsf.sum() - 1) < 1e-14
    delta = d3d.GaussianLineSpreadFunction(0.).as_vector(cube)           # sigma == 0 branch
    assert delta[16] == 1.0 and delta.sum() == 1.0           # centre (D-1)//2 - (D%2-1) = D/2
    v = np.arange(32.)
    assert d3d.VectorLineSpreadFunction(v).as_vector(cube) is v
    np.testing.assert_array_equal(sf.muse_like_lsf_vector(128), O.muse_like_lsf(128))
    with pytest.raises(ImportError):                                      # mpdaf is absent
        d3d.MUSELineSpreadFunction()
    ana = d3d.MUSELineSpreadFunction(model="analytic").as_vector(cube)
    assert ana.shape == (32,) and abs(ana.sum() - 1) < 1e-14 and np.argmax(ana) == 16


def test_line_model_contract():
    m = d3d.SingleGaussianLineModel()
    assert m.parameters() == ['a', 'c', 'w'] and m.gibbs_parameter_index() == 0
    x = np.arange(16)
    np.testing.assert_array_equal(m.modelize(None, x, [2., 7.5, 1.5]),
                                  O.gaussian_line(x, 2., 7.5, 1.5))
    base = d3d.LineModel()
    for call in (base.parameters, lambda: base.min_boundaries(None),
                 lambda: base.max_boundaries(None), lambda: base.modelize(None, x, [1, 2, 3])):
        with pytest.raises(NotImplementedError):
            call()
    assert base.gibbs_parameter_index() is None


def test_median_clip_and_percentile_mask():
    rng = np.random.default_rng(5)
    data = rng.normal(1.0, 2.0, size=(20, 12, 2))
    data[0, 0, 0] = np.nan
    data[5, 5, 1] = 1e3
    assert d3d.median_clip(data.copy(), 2.5) == O.median_clip(data.copy(), 2.5)
    cube = muse_cube((8, 10, 10), seed=3)
    mask = d3d.above_percentile(cube, 60)
    img = cube.data.sum(0)
    assert set(np.unique(mask)) == {0.0, 1.0}
    np.testing.assert_array_equal(mask == 1, img >= np.percentile(img, 60))


def test_fits_round_trip(tmp_path):
    cube = muse_cube((5, 4, 3), seed=9)
    path = str(tmp_path / "c.fits")
    cube.to_fits(path, clobber=True)
    back = d3d.Cube.from_fits(path)
    np.testing.assert_array_equal(back.data, cube.data)
    assert abs(back.get_step(1).to('arcsec').value - 0.2) < 1e-9
    assert abs(back.get_step(0).to('um').value - 1.25e-4) < 1e-15
    with pytest.raises(IOError):
        cube.to_fits(path)                                   # clobber=False


@pytest.mark.skipif(not os.path.isdir(REF_INPUT), reason="reference fixtures not present")
def test_reads_the_reference_fits_fixtures():
    cube = d3d.Cube.from_fits(os.path.join(
        REF_INPUT, "GalPaK_cube_1101_size4.08_flux1e-16_incl60_vmax199_disp80_seeing1.0_PAm50.fits"))
    assert cube.shape == (30, 30, 30)
    assert abs(cube.get_step(1).to('arcsec').value - 0.2) < 1e-6
    assert abs(cube.get_step(0).to('um').value - 1.25e-4) < 1e-12
    img = d3d.MUSE().fsf.as_image(cube)
    assert img.shape == (13, 13)                             # FWHM 1" = 5 px -> ceil(6 sigma) = 13


# ---- Run(): validation happens before any device work ------------------------------

def test_run_rejects_bad_inputs_like_the_reference():
    cube = muse_cube((16, 9, 9))
    inst = d3d.MUSE()
    with pytest.raises(ValueError, match="empty"):          # lib/run.py:135-136
        d3d.Run(d3d.Cube(), inst)
    with pytest.raises(TypeError):                          # lib/run.py:122-134
        d3d.Run(np.zeros((4, 4, 4)), inst)
    with pytest.raises(TypeError, match="Instrument"):      # lib/run.py:203-204
        d3d.Run(cube, instrument="MUSE")
    with pytest.raises(AssertionError):                     # lib/run.py:112-114
        d3d.Run(cube, inst, keep_one_in=0)
    with pytest.raises(AssertionError):
        d3d.Run(cube, inst, max_iterations=0)
    tiny = d3d.MUSE().build_cube(np.full((16, 9, 9), 1e-19))
    with pytest.raises(AssertionError, match="too small"):  # lib/run.py:140-143
        d3d.Run(tiny, inst)
    with pytest.raises(ValueError, match="correct shape"):  # lib/run.py:195-198
        d3d.Run(cube, inst, variance=np.ones((16, 9, 8)))
    with pytest.raises(TypeError, match="variance"):        # lib/run.py:183-184
        d3d.Run(cube, inst, variance=3.0)
    even = d3d.Instrument(d3d.GaussianLineSpreadFunction(1e-4),
                          d3d.ImageFieldSpreadFunction(np.ones((4, 4)) / 16))
    with pytest.raises(ValueError, match="odd"):            # lib/run.py:210-211
        d3d.Run(cube, even)
    with pytest.raises(TypeError, match="LineModel"):       # lib/run.py:231-232
        d3d.Run(cube, inst, model=dict)

    class Inverted(d3d.SingleGaussianLineModel):
        def min_boundaries(self, runner):
            return [5, 0, 0]

        def max_boundaries(self, runner):
            return [1, 10, 10]

    with pytest.raises(ValueError, match="inconsistent"):   # lib/run.py:244-245
        d3d.Run(cube, inst, model=Inverted)
    with pytest.raises(ValueError, match="Initial params"): # lib/run.py:300-305
        d3d.Run(cube, inst, initial_parameters=np.zeros((3, 3, 3)))
    # chains= (additive, like seed=): a positive integer; a checkpoint resumes the number of
    # chains it was written with
    with pytest.raises(AssertionError, match="chains"):
        d3d.Run(cube, inst, chains=0)
    one_chain_state = dict(iteration=3, seed=12345, accepted_count=5, sweep_origin=0)
    with pytest.raises(ValueError, match="holds 1 chain"):
        d3d.Run(cube, inst, chains=2, resume_state=one_chain_state)
    with pytest.raises(ValueError, match="holds 4 chain"):
        d3d.Run(cube, inst, resume_state=dict(one_chain_state, n_chains=4))
    with pytest.raises(ValueError, match="one map per chain"):
        d3d.Run(cube, inst, chains=2, initial_parameters=np.zeros((3, 9, 9, 3)))



def test_bench_refuses_multi_gpu_without_devices():
    """`bench.py --gpus N` without a launcher starts its own ranks -- and must exit
    non-zero, not silently measure one GPU, when the devices are not there."""
    import subprocess
    import sys
    from deconv3d_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("needs a box without HIP devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "HIP device" in r.stderr and not r.stdout.strip()


def test_fits_header_cards_survive_awkward_values(tmp_path):
    """Strings with quotes or longer than a card, and non-finite numbers, in a header
    written by Cube.to_fits (lib/run.py:797-806 writes the run's cubes with the input
    cube's metadata)."""
    from deconv3d_amd.cube import read_fits, write_fits
    data = np.arange(24, dtype=np.float64).reshape(2, 3, 4)
    header = {"OBJECT": "O'Neill's galaxy", "COMMENT1": "x" * 100, "NANVAL": float("nan"),
              "INFVAL": float("inf"), "EXPTIME": 12.5, "NSCANS": 3, "FLAG": True,
              "TRAIL": "ends with quote'"}
    path = str(tmp_path / "c.fits")
    write_fits(path, data, header, clobber=True)
    raw = open(path, "rb").read()
    assert len(raw) % 2880 == 0
    back, h = read_fits(path)
    np.testing.assert_array_equal(back, data)
    assert h["OBJECT"] == "O'Neill's galaxy" and h["TRAIL"] == "ends with quote'"
    assert h["COMMENT1"] == "x" * 68
    assert "NANVAL" not in h and "INFVAL" not in h
    assert h["EXPTIME"] == 12.5 and h["NSCANS"] == 3 and h["FLAG"] is True
    # every card is exactly 80 characters and every string card has its closing quote
    head = raw[:raw.index(b"END" + b" " * 77) + 80].decode("latin1")
    for i in range(0, len(head), 80):
        card = head[i:i + 80]
        if card[8:10] == "= " and card[10:].lstrip().startswith("'"):
            assert card.rstrip().endswith("'"), card

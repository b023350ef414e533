"""CPU-only checks of the host side that mirrors the reference's scripts: flags, file conventions, the
preprocess steps around the resample, the sample helpers of utils.py.  No kernel is launched here."""
import io
import os
import sys
import zipfile

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_str2bool_matches_reference_table():
    """/root/reference/utils.py:23-29: bools pass through, 'true'/'false' in any case, anything else -> None
    (the reference's `in ('true')` is a substring test on a plain string; whole words are what its callers pass)"""
    from cfd_amd.cli import str2bool
    assert str2bool(True) is True and str2bool(False) is False
    for s in ("true", "True", "TRUE"):
        assert str2bool(s) is True
    for s in ("false", "False", "FALSE"):
        assert str2bool(s) is False
    assert str2bool("maybe") is None


@pytest.mark.parametrize("field,default_dir", [("U", "U/training_U/"), ("P", "P/training_P/"),
                                               ("DeltaP", "DeltaP/training_DeltaP/")])
def test_train_flags_and_defaults(field, default_dir):
    """the five flags of U/train.py:28-34 with the reference's defaults, short and long spellings"""
    from cfd_amd.cli import train_parser
    a = train_parser(field).parse_args([])
    assert (a.checkpoint, a.epocheval, a.numepoch, a.trainpath, a.directory) == \
        (False, 1, 10000, "30_case_duyanh/processed/", default_dir)
    a = train_parser(field).parse_args(["-c", "True", "-e", "5", "-n", "7", "-t", "in/", "-d", "out/"])
    assert (a.checkpoint, a.epocheval, a.numepoch, a.trainpath, a.directory) == (True, 5, 7, "in/", "out/")
    b = train_parser(field).parse_args(["--checkpoint", "false", "--epocheval", "5", "--numepoch", "7",
                                        "--trainpath", "in/", "--directory", "out/"])
    assert b.checkpoint is False and (b.epocheval, b.numepoch) == (5, 7)


def test_split_is_the_references_90_10():
    """train_test_split(range(n), test_size=0.1, random_state=42) (U/train.py:53): BASELINE configs[0] = 100 cases"""
    from cfd_amd.cli import _split
    tr, va = _split(100)
    assert len(tr) == 90 and len(va) == 10 and sorted(tr + va) == list(range(100))
    tr2, va2 = _split(100)
    assert tr == tr2 and va == va2                      # deterministic
    assert va[:3] == [83, 53, 70]                       # sklearn's permutation for seed 42 (pinned here)


def test_resume_step_rounds_down_to_an_epoch(tmp_path):
    """steps.txt holds the global step; training resumes at the start of that epoch (U/train.py:93-101)"""
    from cfd_amd.cli import _resume_step
    d = str(tmp_path) + "/"
    assert _resume_step(d, 90, chkpt=True) == 0         # no file yet
    open(d + "steps.txt", "w").write("275\n")
    assert _resume_step(d, 90, chkpt=True) == 270
    assert _resume_step(d, 90, chkpt=False) == 0


def test_generate_samples_shapes_and_labels():
    """utils.py:33-62: real -> [[A, B], ones(n, patch, 8*patch, 1)], fake -> (G(A), zeros(...)), float32 fields"""
    import utils

    class G:                                            # stands in for the model object: predict([x]) -> array
        def predict(self, x):
            assert isinstance(x, list) and len(x) == 1
            return np.full(x[0].shape, 0.25, dtype=np.float64)

    xs = [np.full((64, 512, 1), i, dtype=np.float64) for i in range(5)]
    ys = [np.full((64, 512, 1), -i, dtype=np.float64) for i in range(5)]
    (a, b), y = utils.generate_real_samples(xs, ys, [3, 1], 4)
    assert a.dtype == np.float32 and b.dtype == np.float32 and a.shape == (2, 64, 512, 1)
    assert a[0, 0, 0, 0] == 3 and b[1, 0, 0, 0] == -1
    assert y.shape == (2, 4, 32, 1) and (y == 1).all()
    f, y0 = utils.generate_fake_samples(G(), a, 4)
    assert f.dtype == np.float32 and f.shape == a.shape and (f == 0.25).all()
    assert y0.shape == (2, 4, 32, 1) and (y0 == 0).all()


def _case_csv(n=50, seed=0):
    r = np.random.default_rng(seed)
    df = pd.DataFrame({"x": np.zeros(n), "y": r.uniform(-2e-3, 2e-3, n), "z": r.uniform(0, 5e-2, n),
                       "p": r.normal(size=n), "U_x": r.normal(size=n), "U_y": r.normal(size=n),
                       "U_z": r.normal(size=n)})
    return df


def test_preprocess_steps_around_the_resample(tmp_path):
    """pixels_coords_testing.py:77-145 of the reference: nested case archives -> latest time directory's midBox.csv
    -> Umean + column order -> duplicate rows dropped.  (The resample itself is the GPU tests' business.)"""
    import pixels_coords_testing as pct
    inner_dir = "caseA/postProcessing/boxUniform"
    old, new = _case_csv(seed=1), _case_csv(seed=2)
    new = pd.concat([new, new.iloc[:7]], ignore_index=True)             # 7 duplicate rows
    inner = io.BytesIO()
    with zipfile.ZipFile(inner, "w") as z:
        z.writestr(f"{inner_dir}/100/midBox.csv", old.to_csv(index=False))
        z.writestr(f"{inner_dir}/2500.5/midBox.csv", new.to_csv(index=False))
        z.writestr(f"{inner_dir}/notatime/midBox.csv", "x\n1\n")
    inner_empty = io.BytesIO()
    with zipfile.ZipFile(inner_empty, "w") as z:
        z.writestr("caseB/readme.txt", "no postProcessing here")
    outer = tmp_path / "cases.zip"
    with zipfile.ZipFile(outer, "w") as z:
        z.writestr("caseA.zip", inner.getvalue())
        z.writestr("caseB.zip", inner_empty.getvalue())
    out = tmp_path / "out"
    out.mkdir()
    pct.process_zip_folder(outer, out)
    assert sorted(p.name for p in out.iterdir()) == ["caseA_midBox.csv"]      # caseB skipped, unzipped/ removed
    got = pd.read_csv(out / "caseA_midBox.csv")
    assert len(got) == len(new) and np.array_equal(got["p"].to_numpy(), pd.read_csv(io.StringIO(new.to_csv(index=False)))["p"].to_numpy())

    pct.calculate_umean_and_restructure(out / "caseA_midBox.csv", tmp_path / "r.csv")
    r = pd.read_csv(tmp_path / "r.csv")
    assert list(r.columns) == ["x", "y", "z", "p", "Umean"]
    want = np.sqrt(got["U_x"] ** 2 + got["U_y"] ** 2 + got["U_z"] ** 2).to_numpy()
    assert np.allclose(r["Umean"].to_numpy(), want, rtol=1e-15, atol=0)
    assert len(r.drop_duplicates()) == len(new) - 7

    bad = tmp_path / "bad.zip"
    bad.write_bytes(b"not a zip")
    pct.extract_zip(bad, tmp_path / "nowhere")                          # reported, not raised (:73-74)


def test_read_points_rejects_a_wrong_table(tmp_path):
    import pixels_coords_testing as pct
    p = tmp_path / "c.csv"
    pd.DataFrame({"x": [0.0], "y": [0.0], "z": [0.0]}).to_csv(p, index=False)
    with pytest.raises(ValueError, match="expected 5 columns"):
        pct._read_points(p)


def _pandas_body(a, cols):
    s = io.StringIO()
    pd.DataFrame(a, columns=cols).to_csv(s, index=False)
    txt = s.getvalue()
    return txt[txt.index("\n") + 1:].encode()


def test_csv_writer_is_byte_identical_to_pandas(tmp_path):
    """cfd_csv_format_rows (host-side writer of the resample's result table) against the real pandas `to_csv` the
    reference calls (pixels_coords_testing.py:226): same bytes for every magnitude, the layout switches of repr(float)
    (1e-4 / 1e-5, 1e15 / 1e16), integers, signed zeros, subnormals, extremes, inf and NaN."""
    from cfd_amd import ops
    r = np.random.default_rng(7)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 123456789012345.0, 1e15, 9999999999999998.0,
                        1e16, 1.2345e16, 1e22, 1e23, 1.7976931348623157e308, 2.2250738585072014e-308, 5e-324, 1e-310,
                        0.1, 0.2 + 0.1, 1 / 3, 2 / 3, 100.0, 1072.033, 0.001953125, -0.002 + 0.5 * 4e-3 / 64,
                        np.inf, -np.inf, np.nan, 1e100, 1.5e-100, 123.456, 4503599627370496.0, 9007199254740993.0])
    special = np.concatenate([special, np.full(4 - len(special) % 4, 7.25)]) if len(special) % 4 else special
    mags = 10.0 ** r.integers(-320, 308, size=4000).astype(np.float64)
    with np.errstate(over="ignore", under="ignore"):
        rnd = np.concatenate([r.normal(size=4000) * mags, r.normal(size=2000), r.integers(-10**6, 10**6, 2000).astype(np.float64),
                              np.frombuffer(r.bytes(8 * 4000), dtype=np.float64)])
    a = np.concatenate([special, rnd[: len(rnd) // 4 * 4]]).reshape(-1, 4)
    cols = ["a", "b", "c", "d"]
    got, want = ops.csv_format_rows(a), _pandas_body(a, cols)
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [(i, x, y) for i, (x, y) in enumerate(zip(g, w)) if x != y][:5]
        raise AssertionError(f"{len(g)} vs {len(w)} lines; first differences: {bad}")
    # the preprocess writer end to end: a realistic result table, header included, read back to the same doubles
    import pixels_coords_testing as pct
    tab = np.column_stack([np.repeat(-0.002 + (np.arange(8) + 0.5) * 4e-3 / 8, 16), np.tile((np.arange(16) + 0.5) * 50e-3 / 16, 8),
                           r.uniform(0, 3, 128), r.uniform(0, 2, 128), r.uniform(0, 3000, 128)])
    tab[::5, 2:] = 0.0
    pct._write_table(tab, tmp_path / "t.csv")
    ref = io.StringIO()
    pd.DataFrame(tab, columns=pct.OUT_COLUMNS).to_csv(ref, index=False)
    assert (tmp_path / "t.csv").read_bytes() == ref.getvalue().encode()
    assert ops.csv_format_rows(np.zeros((0, 5))) == b""
    with pytest.raises(ValueError):
        ops.csv_format_rows(np.zeros(5))


def test_read_cases_threaded_equals_serial_in_listing_order(tmp_path, monkeypatch):
    """utils.py:70-87 reads the folder in os.listdir order; the threaded ingest must return the same grids in that order"""
    from cfd_amd import data as D
    r = np.random.default_rng(3)
    folder = tmp_path / "processed"
    folder.mkdir()
    (folder / "a_directory").mkdir()                   # not a file: skipped, as the reference's isfile test does
    for i in range(9):
        a = r.normal(size=(8 * 16, 5))
        pd.DataFrame(a, columns=["y_center", "z_center", "p_center", "Umean_center", "p_convert"]).to_csv(
            folder / f"case_{(i * 7) % 9}.csv", index=False)
    monkeypatch.setattr(D, "INGEST_THREADS", 1)
    ps1, us1 = D.read_cases(str(folder), 8, 16)
    monkeypatch.setattr(D, "INGEST_THREADS", 4)
    ps4, us4 = D.read_cases(str(folder), 8, 16)
    assert len(ps1) == len(ps4) == 9
    assert all(np.array_equal(a, b) for a, b in zip(ps1, ps4)) and all(np.array_equal(a, b) for a, b in zip(us1, us4))
    names = [n for n in os.listdir(folder) if os.path.isfile(folder / n)]
    first = _dictreader_column(folder / names[0], "p_convert").reshape(8, 16)
    assert np.array_equal(ps4[0], first)


def _dictreader_column(path, col):
    """the reference's parse, restated: utils.py:72-83 (csv.DictReader, float(row[col]))"""
    import csv
    with open(path, "r") as fh:
        return np.asarray([float(row[col]) for row in csv.DictReader(fh)], dtype=np.float64)


@pytest.mark.parametrize("native", ["1", "0"])
def test_read_case_parses_like_python_float(tmp_path, native, monkeypatch):
    """A8: utils.py:82-83 parses with float(); pandas' default parser differs from it in the last bit for ~15 % of
    17-digit values, so the ingest uses the library's strtod reader (native = 1) or pandas' round-trip parser (0, and
    the fallback).  200 000 values, bit for bit."""
    from cfd_amd import data as D
    monkeypatch.setenv("CFD_AMD_NATIVE_CSV", native)
    r = np.random.default_rng(11)
    h, w = 200, 500
    a = np.concatenate([r.normal(size=(h * w // 2, 2)) * 10.0 ** r.integers(-8, 8, size=(h * w // 2, 1)),
                        r.uniform(0, 3000, size=(h * w - h * w // 2, 2))])
    path = tmp_path / "case.csv"
    with open(path, "w") as fh:
        fh.write("y_center,z_center,p_center,Umean_center,p_convert\n")
        for u, p in a:
            fh.write("0.0,0.0,0.0,%s,%s\n" % (repr(float(u)), "%.17g" % p))   # shortest and 17-digit spellings
    p_got, u_got = D.read_case(str(path), h, w)
    p_ref, u_ref = _dictreader_column(path, "p_convert"), _dictreader_column(path, "Umean_center")
    assert np.array_equal(p_got.ravel().view(np.uint64), p_ref.view(np.uint64))
    assert np.array_equal(u_got.ravel().view(np.uint64), u_ref.view(np.uint64))
    default = pd.read_csv(path)["p_convert"].to_numpy()
    assert (default.view(np.uint64) != p_ref.view(np.uint64)).any(), "the default parser no longer differs: test is vacuous"


def test_native_csv_reader_spellings_and_fallbacks(tmp_path):
    """cfd_csv_parse_columns against csv.DictReader + float() on the spellings float() accepts, and its refusals (-2:
    the ingest then takes pandas' round-trip parser): quotes, '_' separators, hex, empty fields, ragged rows, a missing
    column, more rows than expected."""
    import ctypes as C
    from cfd_amd import _lib, data as D
    lib = _lib.load()
    vals = ["nan", "inf", "-inf", "Infinity", "-NaN", "1e-320", "4.9e-324", "1.7976931348623157e308", "+1.5", " 2.5 ", "1E5",
            ".5", "5.", "-0.0", "0.1", "123456789012345678901234567890", "2.2250738585072011e-308", "9007199254740993",
            "0.30000000000000004", "1e23", "8.41e21", "2.4703282292062327e-324", "1e400", "-1e-400"]
    text = "a,p_convert,b,Umean_center\r\n" + "".join(f"0,{v},7,{w}\r\n" for v, w in zip(vals, reversed(vals))) + "\n"
    path = tmp_path / "s.csv"
    path.write_bytes(text.encode())
    want_p, want_u = _dictreader_column(path, "p_convert"), _dictreader_column(path, "Umean_center")
    raw = path.read_bytes()
    pa, ua = np.full(len(vals), 7.0), np.full(len(vals), 7.0)
    n = lib.cfd_csv_parse_columns(raw, len(raw), b"p_convert", b"Umean_center", pa.ctypes.data, ua.ctypes.data, len(vals))
    assert n == len(vals)
    assert np.array_equal(pa.view(np.uint64), want_p.view(np.uint64)) and np.array_equal(ua.view(np.uint64), want_u.view(np.uint64))
    head = "p_convert,Umean_center\n"
    for bad in (head + '"1.0",2.0\n', head + "1_0,2.0\n", head + "0x1p3,2.0\n", head + ",2.0\n", head + "1.0\n",
                head + "1.0,2.0,3.0\n", "x,Umean_center\n1.0,2.0\n", head + "1.0,2.0\n3.0,4.0\n", head + "1.0.0,2\n",
                head + "1e,2\n", "no newline"):
        b = bad.encode()
        assert lib.cfd_csv_parse_columns(b, len(b), b"p_convert", b"Umean_center", pa.ctypes.data, ua.ctypes.data, 1) == -2, bad
    # the ingest falls back to pandas for a file the native reader refuses (quoted numbers), with the same doubles
    q = tmp_path / "q.csv"
    q.write_text("p_convert,Umean_center\n" + "".join(f'"{0.1 * i!r}",{0.3 * i!r}\n' for i in range(6)))
    p, u = D.read_case(str(q), 2, 3)
    assert np.array_equal(p.ravel(), np.array([0.1 * i for i in range(6)])) and np.array_equal(u.ravel(), np.array([0.3 * i for i in range(6)]))


def test_forward_chunked_host_logic(monkeypatch):
    """keras_api._forward_chunked with a stand-in forward on CPU tensors: chunk boundaries, the ragged tail, the
    sample0 offsets handed to the forward, and that each chunk's result is copied out of the (reused) buffer."""
    import torch
    from cfd_amd import keras_api as K
    calls = []
    scratch = torch.empty(4, 3)                      # like the nets: every forward returns the same reused buffer

    def forward(x, seed=0, sample0=0):
        calls.append((x.shape[0], seed, sample0))
        scratch[:x.shape[0]] = x * 2 + sample0
        return scratch[:x.shape[0]]

    x = torch.arange(30, dtype=torch.float32).view(10, 3)
    monkeypatch.setenv("CFD_AMD_PREDICT_CHUNK", "4")
    y = K._forward_chunked(forward, x, True, seed=9)
    assert calls == [(4, 9, 0), (4, 9, 4), (2, 9, 8)]
    off = torch.tensor([0] * 4 + [4] * 4 + [8] * 2, dtype=torch.float32).view(10, 1)
    assert torch.equal(y, x * 2 + off)
    calls.clear()
    y = K._forward_chunked(lambda x, training=False: (calls.append(x.shape[0]), x + 1)[1], x, False, training=False)
    assert calls == [4, 4, 2] and torch.equal(y, x + 1)
    calls.clear()
    monkeypatch.setenv("CFD_AMD_PREDICT_CHUNK", "16")     # one pass: the forward's own buffer is returned as is
    assert K._forward_chunked(forward, x[:3], True, seed=1) is not None and calls == [(3, 1, 0)]


def test_eval_plans_select_and_scale_like_the_six_reference_scripts():
    """each of U|P|DeltaP/{valid,test}.py scores its own set of cases with its own scale.
    Restated here from the reference scripts (file:line in cli.EVAL_PLANS) on a toy folder of 20 cases."""
    from cfd_amd import cli
    n = 20
    r = np.random.default_rng(5)
    p_grids = [r.uniform(0, 100 * (i + 1), size=(4, 8)) for i in range(n)]       # case i's maximum grows with i
    dps = [float(10 * (i + 1) + 0.123456789) for i in range(n)]
    train_idx, valid_idx = np.array([3, 0, 17, 5, 9]), np.array([19, 2])
    # U/valid.py:44-51: train_size = int(0.9 n + 1); the last n - train_size cases; no scale
    assert cli.eval_selection("U", "valid", n) == [19]
    assert cli.eval_selection("U", "valid", 100) == list(range(91, 100))
    assert cli.eval_scale("U", "valid", None) == 1.0
    with pytest.raises(ValueError):
        cli.eval_selection("U", "valid", 5)
    # U/test.py:77-86: train_size = 0 -> every case
    assert cli.eval_selection("U", "test", n) == list(range(n)) and cli.eval_scale("U", "test", None) == 1.0
    # P/valid.py:45-66,95: valid_indices.txt cases, 2 * max over the train_indices.txt cases of the same folder
    assert cli.eval_selection("P", "valid", n, train_idx, valid_idx) == [19, 2]
    want = 2.0 * max(p_grids[i].max() for i in train_idx)
    assert cli.eval_scale("P", "valid", p_grids, train_idx) == want
    # P/test.py:76-90,123-125,168: every case of the test folder, 2 * P_max of the TRAINING folder's train cases
    assert cli.eval_selection("P", "test", 7, train_idx, None) == list(range(7))
    assert cli.eval_scale("P", "test", p_grids, train_idx) == want
    # DeltaP/valid.py:81-126,157: valid_indices.txt cases, P_max = float32 max of the train cases' DeltaP, factor 1
    assert cli.eval_selection("DeltaP", "valid", n, train_idx, valid_idx) == [19, 2]
    assert cli.eval_scale("DeltaP", "valid", dps, train_idx) == float(np.float32(dps[17]))
    # DeltaP/test.py (a copy of P/test.py): every case, factor 2
    assert cli.eval_selection("DeltaP", "test", 7, train_idx, None) == list(range(7))
    assert cli.eval_scale("DeltaP", "test", dps, train_idx) == 2.0 * dps[17]
    # indices that do not belong to the folder are refused, as is a missing index file
    with pytest.raises(ValueError):
        cli.eval_selection("P", "valid", 10, train_idx, valid_idx)
    with pytest.raises(ValueError):
        cli.eval_selection("P", "valid", n, train_idx, None)
    with pytest.raises(ValueError):
        cli.eval_scale("P", "test", p_grids[:10], train_idx)
    # a one-case valid_indices.txt loads as a 0-d array (np.loadtxt)
    assert cli.eval_selection("DeltaP", "valid", n, train_idx, np.array(4)) == [4]
    # every plan names the folder its script hard-codes
    assert cli.EVAL_PLANS[("P", "valid")]["folder"] == "./100_case_si_test/processed/"
    assert cli.EVAL_PLANS[("DeltaP", "valid")]["folder"] == "./130_case/processed/"
    assert cli.EVAL_PLANS[("U", "test")]["folder"] == "./30_case_duyanh/processed/"


# ---- bench.py: what it did not measure is named, and a failure or hang is never a green run -------------------------
def _bench_dry(extra_env, timeout=120):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CFD_BENCH_DRY="1", **extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                          text=True, timeout=timeout, cwd=root)


def _last_json_line(text):
    import json
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_bench_extras_error_under_two_ranks_exits_nonzero():
    """WORLD_SIZE = 2 (gloo, CFD_BENCH_DRY: canned workloads, one collective each): an extra workload that raises is
    recorded in the line AND the run exits non-zero; the spawner stops the other rank and relays the code"""
    r = _bench_dry({"CFD_BENCH_FAIL_EXTRA": "train_u_bf16"})
    assert r.returncode == 4, (r.returncode, r.stderr[-800:])
    line = _last_json_line(r.stdout)
    assert "forced by CFD_BENCH_FAIL_EXTRA" in line["also"]["train_u_bf16"]["error"]
    assert "infer_all_bf16" in line["also"] and "resample" not in line["also"]      # measured before / never started
    assert "train_u_bf16" in r.stderr


def test_bench_hang_in_a_collective_exits_nonzero_and_names_the_workload():
    """one rank never reaches an extra's collective: the watchdog prints what was measured, names the workload in
    flight and leaves with code 3 (it used to leave with 0)"""
    r = _bench_dry({"CFD_BENCH_HANG_EXTRA": "train_deltap_bf16", "CFD_BENCH_WATCHDOG_S": "4"})
    assert r.returncode == 3, (r.returncode, r.stderr[-800:])
    line = _last_json_line(r.stdout)
    assert "train_deltap_bf16" in line["watchdog"] and "exit code 3" in line["watchdog"]
    assert "train_u_fp32_b1" in line["also"] and "train_deltap_bf16" not in line["also"]


def test_bench_green_run_two_ranks_exits_zero_and_has_configs0_key():
    r = _bench_dry({})
    assert r.returncode == 0, r.stderr[-800:]
    line = _last_json_line(r.stdout)
    assert line["n_gpus"] == 2 and set(line["also"]) == {"infer_all_bf16", "train_u_bf16", "train_u_bf16_b32",
                                                         "train_u_fp32_b1", "train_deltap_bf16", "resample"}


def test_bench_names_the_profile_its_pmc_figures_come_from():
    """roofline.traffic / mfma_busy_frac_pmc are read from a committed profile, not measured in the run: the line says
    which file, which commit and whether the kernel sources are the ones running"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    traffic, busy, src = bench.pmc_traffic("infer_all_bf16")
    assert traffic and src and src["file"].startswith("profiles/") and src["file"].endswith("_infer_all_bf16_summary.json")
    assert {"git_head", "csrc_sha16", "same_sources_as_this_run", "note"} <= set(src)
    assert "not measured in this run" in src["note"]
    tr, rsrc = bench.pmc_traffic_resample()
    assert tr and rsrc["file"].endswith("_resample_summary.json")
    assert len(bench.csrc_sha16()) == 16

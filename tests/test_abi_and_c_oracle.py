"""CPU tests: the C-ABI library loads and exports every symbol include/lipmpc.h declares (no compute
without a GPU); the Python struct mirror matches the C struct; the C oracle equals the numpy oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import lipmpc
import lipmpc_oracle as O
import c_oracle
from helpers import closed_loop_problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "lipmpc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lipmpc_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = lipmpc._lib.load()
    names = _declared_functions()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lipmpc.h but not exported"
    assert set(names) == set(lipmpc._lib.EXPORTS)
    assert lib.lipmpc_version() == lipmpc._lib.ABI_VERSION == 5
    assert b"ok" == lib.lipmpc_strerror(0)


def test_loader_refuses_another_abi(tmp_path):
    """_lib.load() binds its argument lists for ONE ABI version: a library of another version (a stale build, a historical
    one through LIPMPC_LIB) or an instrumented variant (version + LIPMPC_VARIANT_BASE: other buffer shapes) must be refused
    before any pointer is handed over.  Stub libraries that only export lipmpc_version()."""
    import sys
    for ver, allow, ok in ((4, "0", False), (6, "0", False), (1005, "0", False), (1005, "1", True), (1004, "1", False)):
        src = tmp_path / f"v{ver}.c"
        src.write_text(f"int lipmpc_version(void) {{ return {ver}; }}\n")
        so = tmp_path / f"libv{ver}.so"
        subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
        code = ("import sys; sys.path.insert(0, %r); import lipmpc\n"
                "try:\n    lipmpc._lib.load(); print('LOADED')\n"
                "except RuntimeError as e:\n    print('REFUSED', e)\n"
                "except AttributeError as e:\n    print('LOADED-THEN', e)\n") % ROOT
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LIPMPC_LIB=str(so), LIPMPC_ALLOW_VARIANT=allow),
                           capture_output=True, text=True, timeout=120)
        out = r.stdout.strip()
        if ok:      # past the version gate (the stub then lacks the other symbols)
            assert out.startswith("LOADED"), (ver, allow, out, r.stderr[-500:])
        else:
            assert out.startswith("REFUSED") and f"lipmpc_version() = {ver}" in out, (ver, allow, out, r.stderr[-500:])


def test_params_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lipmpc.h"\n'
                   'int main(){printf("%zu %zu %zu %zu", sizeof(lipmpc_params), offsetof(lipmpc_params, dt),'
                   ' offsetof(lipmpc_params, omega_max), offsetof(lipmpc_params, k0_tol));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    size, o_dt, o_om, o_k0 = map(int, subprocess.check_output([str(exe)]).split())
    S = lipmpc._lib.LipmpcParamsC
    assert (size, o_dt, o_om, o_k0) == (C.sizeof(S), S.dt.offset, S.omega_max.offset, S.k0_tol.offset)


def test_default_params_are_the_reference_config():
    lib = lipmpc._lib.load()
    p = lipmpc._lib.LipmpcParamsC()
    assert lib.lipmpc_default_params(C.byref(p)) == 0
    d = lipmpc.LipMpcParams()
    # config.yml:2-17, HumanoidMpc.py:20-22, :200
    assert (p.dt, p.g, p.h_com, p.alpha, p.ell) == (0.4, 9.81, 1.0, 3.6, 0.05) == (d.dt, d.g, d.h_com, d.alpha, d.ell)
    assert tuple(p.l_max) == (0.1, 0.1) and tuple(p.l_min) == (-0.1, -0.1)
    assert tuple(p.v_min) == (-0.1, 0.1) and tuple(p.v_max_xy) == (0.8, 0.4)
    assert abs(p.omega_max - 0.156 * np.pi) < 1e-16
    q = d.to_c()
    assert lib.lipmpc_num_rows(C.byref(q)) == d.num_rows == 27
    q.N, q.n_obs_max = 8, 10
    assert lib.lipmpc_num_rows(C.byref(q)) == 162 and lib.lipmpc_active_words(C.byref(q)) == 3
    # schedule buffer of the LiDAR front end: header + start order + reading counts (include/lipmpc.h)
    assert lib.lipmpc_lidar_schedule_words(4096) == 2 + 2 * 4096 and lib.lipmpc_lidar_schedule_words(0) == 2
    assert lib.lipmpc_lidar_schedule_words(-1) < 0
    # split-launch workspace: argument errors without a handle
    assert lib.lipmpc_workspace_bytes(C.c_void_p(0), 4096) < 0 and lib.lipmpc_set_workspace(C.c_void_p(0), C.c_void_p(0), 0) == -1
    # argument errors of the LiDAR entry points never reach a device (no GPU needed)
    z = C.c_void_p(0)
    assert lib.lipmpc_lidar_c_eta_batch(0, 1, 360, 0, 1, 1, C.c_double(1.5), C.c_double(0.3), 3, 12, 32, *([z] * 14)) == -1   # no c_eta
    assert lib.lipmpc_sense_plan_step_batch(z, 1, 360, 0, 1, 1, C.c_double(1.5), C.c_double(0.3), 3, *([z] * 24)) == -1       # no handle
    one = C.c_void_p(8)                                                                                                        # (never dereferenced)
    assert lib.lipmpc_lidar_c_eta_batch(0, 1, 360, 70000, 5, 1, C.c_double(1.5), C.c_double(0.3), 3, 12, 32, *([one] * 14)) == -2  # obstacle indices are 16 bits in LDS


def test_create_without_gpu_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = lipmpc._lib.load()
    p = lipmpc.LipMpcParams().to_c()
    h = C.c_void_p()
    assert lib.lipmpc_create(C.byref(p), 0, C.byref(h)) == -3          # LIPMPC_E_HIP, no crash
    with pytest.raises(RuntimeError, match="no CPU path"):
        lipmpc.BatchedLipMpc(lipmpc.LipMpcParams())


@pytest.mark.parametrize("N,n_obs", [(3, 3), (8, 10)])
def test_c_oracle_equals_numpy_oracle(N, n_obs):
    probs = list(closed_loop_problems(N, n_obs, 3, 15, seed=40 + N))
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    out = c_oracle.plan_step_batch(P, np.array([p[0] for p in probs]), np.array([p[1] for p in probs], float),
                                   np.array([p[2] for p in probs], np.int8), xy, nv,
                                   np.array([p[4] for p in probs], float), n_threads=2)
    act = lipmpc.unpack_active(out["active"], P.num_rows)
    work = lipmpc.unpack_active(out["working"], P.num_rows)
    for b, (st, goal, s0, obs, delta) in enumerate(probs):
        r = O.plan_step(st, goal, s0, obs, delta, O.Params(N=N))
        assert r["status"] == out["status"][b] and r["iters"] == out["iters"][b]
        assert np.max(np.abs(out["U"][b] - r["U"])) < 1e-8
        assert np.max(np.abs(out["X"][b] - r["X"])) < 1e-8
        assert np.array_equal(out["theta"][b], r["theta"]) and np.array_equal(out["omega"][b], r["omega"])
        assert np.array_equal(out["c_eta"][b][:, :2], r["c"]) and np.array_equal(out["c_eta"][b][:, 2:], r["eta"])
        assert np.array_equal(act[b], r["active"]) and np.array_equal(work[b], r["working"])
        if r["status"] == 0:
            # the working set is a subset of the tight set up to the tolerance the certificate holds its rows to (1e-9 << 1e-7)
            assert not np.any(r["working"] & ~r["active"])
            assert abs(out["diag"][b][4] - r["tight_margin"]) <= 1e-9 + 1e-6 * r["tight_margin"]


def test_presolve_keeps_the_optimum_and_both_oracles_agree():
    """The presolve (rows the leg-reach rows make redundant leave the problem, a ballast row keeps their averaging effect on
    the interior-point iteration: oracle presolve_ldcbf) changes the path, not the answer: same statuses, same footsteps to
    1e-7, same active sets (tight sets of the optimum: every problem but those with a row within the answers' distance of
    the tolerance; working sets wherever the certificates are decisive), with and without it; numpy and C oracle agree on both."""
    from helpers import assert_active_sets, compare_active_sets
    N, n_obs = 8, 10
    probs = list(closed_loop_problems(N, n_obs, 4, 12, seed=5))
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    args = (np.array([p[0] for p in probs]), np.array([p[1] for p in probs], float), np.array([p[2] for p in probs], np.int8), xy, nv,
            np.array([p[4] for p in probs], float))
    on = c_oracle.plan_step_batch(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5), *args, n_threads=2)
    off = c_oracle.plan_step_batch(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=lipmpc.FLAG_NO_PRESOLVE), *args, n_threads=2)
    assert np.array_equal(on["status"], off["status"])
    ok = on["status"] == 0
    assert ok.sum() >= 0.9 * len(probs) and np.max(np.abs(on["U"][ok] - off["U"][ok])) < 1e-7
    info, _ = compare_active_sets(ok, on, off)
    assert_active_sets("presolve on / off", info, 0.97)
    assert info["working_mismatch_decisive"] == 0, info
    dropped = 0
    for b, (st, goal, s0, obs, delta) in enumerate(probs):
        x0 = np.asarray(st[:4], float)
        cs, etas, _ = O.list_c_and_eta(x0, obs)
        red, n_d, s_bar = O.presolve_ldcbf(x0, cs, etas, delta, O.Params(N=N))
        dropped += n_d
        assert n_d == red.sum() and (n_d == 0 or s_bar > O.reach_step(O.Params(N=N)))
        for pre in (True, False):
            r = O.plan_step(st, goal, s0, obs, delta, O.Params(N=N, presolve=pre))
            ref = on if pre else off
            assert r["status"] == ref["status"][b] and r["iters"] == ref["iters"][b]
            if r["status"] == 0:
                assert np.max(np.abs(r["U"] - ref["U"][b])) < 1e-8
                # a dropped row is never in the active set
                act = np.asarray(r["active"])[9 * N + n_obs:].reshape(N, n_obs)
                assert not np.any(act & red) if pre else True
    assert dropped > 0.5 * N * n_obs * len(probs)          # most LDCBF rows of these fields are redundant


@pytest.mark.parametrize("cap", [1, 2])
def test_finish_rounds_cap_c_oracle_equals_numpy_oracle(cap):
    """lipmpc_params.finish_rounds reaches both oracles the same way (statuses, rounds and answers agree under a
    tight cap, where part of the batch ends UNCERTIFIED with the interior-point answer)."""
    N, n_obs = 8, 10
    probs = list(closed_loop_problems(N, n_obs, 3, 12, seed=77))
    # (every LDCBF row kept: with the presolve on, this small batch certifies in one round throughout)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, finish_rounds=cap, flags=lipmpc.FLAG_NO_PRESOLVE)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    out = c_oracle.plan_step_batch(P, np.array([p[0] for p in probs]), np.array([p[1] for p in probs], float),
                                   np.array([p[2] for p in probs], np.int8), xy, nv,
                                   np.array([p[4] for p in probs], float), n_threads=2)
    n4 = 0
    for b, (st, goal, s0, obs, delta) in enumerate(probs):
        r = O.plan_step(st, goal, s0, obs, delta, O.Params(N=N, finish_rounds=cap, presolve=False))
        assert r["status"] == out["status"][b]
        if r["status"] in (O.STATUS_SOLVED, O.STATUS_UNCERTIFIED):
            assert r["rounds"] <= cap and out["diag"][b, 0] == r["rounds"]
            assert np.max(np.abs(out["U"][b] - r["U"])) < 1e-8
            n4 += r["status"] == O.STATUS_UNCERTIFIED
    assert cap > 1 or n4 > 0


def test_c_oracle_geometry_against_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "geometry_golden.npz"))
    B = len(d["pts"])
    P = lipmpc.LipMpcParams(N=3, n_obs_max=1, v_max=24)
    st = np.zeros((B, 5)); st[:, 0] = d["pts"][:, 0]; st[:, 2] = d["pts"][:, 1]
    xy = d["rings"][d["which"]][:, None, :, :]
    nv = d["nv"][d["which"]][:, None].astype(np.int32)
    out = c_oracle.plan_step_batch(P, st, st[:, [0, 2]] + 5.0, np.ones(B, np.int8), xy, nv, None, n_threads=4)
    ce = out["c_eta"][:, 0]
    assert np.max(np.abs(ce[:, :2] - d["c"])) <= 4e-15
    h0 = np.sum(ce[:, 2:] * (d["pts"] - ce[:, :2]), axis=1)
    assert np.array_equal(h0 < 0, d["inside"])


def test_synthetic_fields_follow_the_generator_rules():
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    xy, nv = synth.synthetic_fields(16, 10, 0.5, 9.5, (0, 0), (10, 10), seed=3)
    assert xy.shape == (16, 10, 5, 2) and np.all((nv >= 3) & (nv <= 5))
    for b in range(16):
        polys = [xy[b, j, : nv[b, j]] for j in range(10)]
        for i, p in enumerate(polys):
            a, bb = p, np.roll(p, -1, axis=0)
            assert np.all((bb[:, 0] - a[:, 0]) * (np.roll(bb, -1, 0)[:, 1] - a[:, 1])
                          - (bb[:, 1] - a[:, 1]) * (np.roll(bb, -1, 0)[:, 0] - a[:, 0]) > 0)     # strictly convex, CCW
            for q in polys[:i]:
                assert not synth._sat_intersect(p, q)


@pytest.mark.parametrize("name,n_obs,hi,goal", [("fields_cfg2", 10, 9.5, (10.0, 10.0)), ("fields_cfg4", 50, 15.5, (16.0, 16.0))])
def test_synthetic_fields_match_the_reference_generator_in_distribution(golden_dir, name, n_obs, hi, goal):
    """bench.py's obstacle fields come from synth.synthetic_fields, a restatement of the reference's generate_obstacles
    (Utils/obstacles.py:167-206) on numpy's generator; the committed fixtures fields_cfg2 / fields_cfg4 were produced by the
    imported reference itself (tests/golden/make_golden.py).  Same number of fields, two-sample checks of everything the solve
    can see of a field's shape: polygons per field, vertex-count shares, polygon areas, nearest-neighbour distances between
    polygon centres, centre coordinates.  (The solver-level twin -- kept rows after the presolve, iteration counts on both
    sets under the same walk recipe -- is tests/test_gpu_configs.py::test_bench_inputs_match_the_reference_fields.)"""
    from importlib import import_module
    from scipy import stats
    from helpers import field_features
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    B = len(d["nv"])
    xy, nv = synth.synthetic_fields(B, n_obs, 0.5, hi, (0.0, 0.0), goal, seed=1234)
    fr, fs = field_features(d["rings"], d["nv"]), field_features(xy, nv)
    assert np.array_equal(fr["per_field"], fs["per_field"])                    # every field full
    assert np.max(np.abs(fr["vertex_share"] - fs["vertex_share"])) < 0.04, (fr["vertex_share"], fs["vertex_share"])
    for key in ("area", "nearest"):
        p = stats.ks_2samp(fr[key], fs[key]).pvalue
        assert p > 0.01, (key, p, fr[key].mean(), fs[key].mean())
        assert abs(fr[key].mean() - fs[key].mean()) < 0.05 * fr[key].mean(), key
    for c in (0, 1):
        assert stats.ks_2samp(fr["centres"][:, c], fs["centres"][:, c]).pvalue > 0.01


@pytest.mark.timeout(900)
def test_headline_kernel_resource_report():
    """Tripwire for the failure class of round 1 (a kernel reading a register it never wrote, under hundreds of SGPR spills):
    the compiler's resource report of the headline instantiation (plan_step_kernel<16,5,16,*>: BASELINE config 2) must show no
    scratch, no VGPR spill to memory and SGPR spills within the recorded bound; and the solver bodies of the 32-lane split
    launch (solve_list_kernel<32,{1,2,4,13,25},32>: BASELINE config 4), each compiled as its own kernel precisely so that it
    keeps its own register allocation, must be scratch-free too (the single 32-lane dispatching kernel is not: 304 B per lane).
    hipcc cross-compiles without a GPU."""
    import re
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "humanoid-navigation-using-mpc-ldcbf_amd", "csrc", "lipmpc_inst.hip")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DINST_G=16", "-DINST_NL=5", "-DINST_NV=16",
                        "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for blk in blocks:
        name = blk.split()[0]
        if "plan_step_kernel" not in name:
            continue
        seen += 1
        val = lambda key: int(re.search(key + r"[^:]*: (\d+)", blk).group(1))
        assert val("ScratchSize") == 0, (name, val("ScratchSize"))
        assert val("SGPRs Spill") <= 120, (name, val("SGPRs Spill"))
        assert val("Occupancy") == 1
    assert seen == 2                      # the dispatching kernel and the plain one
    for nl in (1, 2, 4, 13, 25):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DINST_LIST", "-DINST_G=32", f"-DINST_NL={nl}",
                            "-DINST_NV=32", "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        blk = [b for b in re.split(r"remark: Function Name: ", r.stderr)[1:] if "solve_list_kernel" in b.split()[0]]
        assert len(blk) == 1
        val = lambda key: int(re.search(key + r"[^:]*: (\d+)", blk[0]).group(1))
        assert val("ScratchSize") == 0 and val("Occupancy") == 1, (nl, val("ScratchSize"))
        assert val("SGPRs Spill") <= 160, (nl, val("SGPRs Spill"))


def test_lidar_kernel_resource_report():
    """The scan's occupancy is what its resource report says it is: 128 registers and at most 10 KB of LDS per wave = 16
    waves per compute unit (a whole batch of 4096 robots resident at once, which the launch order of lidar_order_kernel is
    built on), no scratch, no register spilled to memory."""
    import re
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "humanoid-navigation-using-mpc-ldcbf_amd", "csrc", "lipmpc_lidar.hip")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    blk = [b for b in re.split(r"remark: Function Name: ", r.stderr)[1:] if "lidar_sense_kernel" in b.split()[0]]
    assert len(blk) == 1
    val = lambda key: int(re.search(key + r"[^:]*: (\d+)", blk[0]).group(1))
    assert val("ScratchSize") == 0 and val("VGPRs Spill") == 0
    assert val(" VGPRs") <= 128 and val("AGPRs") == 0 and val("Occupancy") == 4
    assert val("LDS Size") <= 10240, val("LDS Size")              # 160 KB / 16 waves


def test_dev_variants_compile():
    """Every development switch of the kernel sources (tools/variants.txt: the phase-timing instrumentation) still compiles --
    none is part of the product build, each is one `tools/build_variant.sh --check-all` line."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "--check-all"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    names = [ln.split()[0] for ln in open(os.path.join(ROOT, "tools", "variants.txt")) if ln.strip() and not ln.startswith("#")]
    assert len(names) >= 1 and all(f"variant {n} " in r.stdout for n in names), r.stdout

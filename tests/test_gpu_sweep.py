"""Rate-distortion sweep as a product entry point (reference generate_rd.sh:1-17): one job, the
cluster labels and the conditional counters made once, codebooks + quantizer + coding per operating point - through the C-ABI
(qvz_encode_file_sweep, include/qvz_encode.h) and through the qvz binary (QVZ_SWEEP). Every point is pinned to what the
REFERENCE wrote at that point (tests/golden/manifest.json "sweeps": one run of oracle/_ref/qvz_ref_fixedseed per point, md5 of
its .q and -u and the fields of its -s line), not to another encode of this library."""
import os
import re
import subprocess

import pytest

import golden_util as G
import oracle_lib as O
from test_gpu_encode import QVZ_BIN, fresh_process_rand

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")


def write_sweep_input(tmp_path, name):
    rows = G.sweep_rows(name)
    p = tmp_path / (name + ".txt")
    p.write_bytes(rows.tobytes())
    return str(p), rows


@pytest.mark.parametrize("name", sorted(G.SWEEPS))
def test_sweep_points_are_the_reference_files(tmp_path, name):
    """All twenty points of the reference script's sweep (and of its commented -c 3 -r variant): .q and -u md5, stream size,
    rate and distortion as the reference printed them."""
    from qvz_amd import host
    e = G.SWEEPS[name]
    inp, rows = write_sweep_input(tmp_path, name)
    k, mode, ratios = G.sweep_opts(name)
    hmode = host.MODE_RATIO if mode == O.MODE_RATIO else host.MODE_FIXED
    out, u = str(tmp_path / "rd.q"), str(tmp_path / "rd.u")
    fresh_process_rand()
    res = host.encode_file_sweep(inp, out, [(hmode, r) for r in ratios], clusters=k, dist="M", u_path=u, seed_mode=host.SEED_DEBUG)
    assert len(res) == 20
    for i, (pt, r) in enumerate(zip(e["points"], res)):
        qb, ub = open(f"{out}.{i:02d}", "rb").read(), open(f"{u}.{i:02d}", "rb").read()
        assert (len(qb), G.md5(qb), G.md5(ub)) == (pt["q_size"], pt["q_md5"], pt["u_md5"]), f"point {pt['comp']}"
        assert (r.stream_bytes, r.file_bytes, r.lines, r.columns) == (pt["size"], pt["q_size"], rows.shape[0], rows.shape[1] - 1)
        assert f"{r.stream_bytes * 8.0 / (r.lines * r.columns):.4f}" == pt["rate"] and f"{r.distortion:.4f}" == pt["distortion"]
    assert res[0].t_cluster > 0 and res[0].t_stats > 0                         # the shared phases count for the first point ...
    assert all(r.t_cluster == 0 and r.t_stats == 0 and r.t_upload == 0 for r in res[1:])   # ... and are not run again


def test_cli_sweep_prints_the_lines_generate_rd_collects(tmp_path):
    """QVZ_SWEEP=rd through the binary: twenty -s lines whose rate / distortion / size fields are the reference's (what
    generate_rd.sh tees into its stats file and awk cuts, :8,16), outputs <out>.00 ... <out>.19 with the reference's md5;
    QVZ_SWEEP_OUT=last leaves only <out>, the last point's file, as the script's twenty encodes of one name do."""
    name = "rd_t2k"
    e = G.SWEEPS[name]
    inp, _ = write_sweep_input(tmp_path, name)
    out = str(tmp_path / "o.q")
    env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SWEEP="rd")
    txt = subprocess.run([QVZ_BIN, "-c", "1", inp, out], env=env, capture_output=True, text=True, check=True).stdout
    lines = [ln for ln in txt.splitlines() if ln.startswith("rate,")]
    assert len(lines) == 20
    for i, (ln, pt) in enumerate(zip(lines, e["points"])):
        m = re.fullmatch(r"rate, ([\d.]+), distortion, ([\d.]+), time, [\d.]+, size, (\d+) ", ln)
        assert m and (m.group(1), m.group(2), int(m.group(3))) == (pt["rate"], pt["distortion"], pt["size"]), ln
        assert G.md5(open(f"{out}.{i:02d}", "rb").read()) == pt["q_md5"]
    assert not os.path.exists(out)
    subprocess.run([QVZ_BIN, "-c", "1", inp, out], env=dict(env, QVZ_SWEEP_OUT="last"), capture_output=True, check=True)
    assert G.md5(open(out, "rb").read()) == e["points"][-1]["q_md5"] and not os.path.exists(out + ".20")
    # an explicit list, spelled like the script's commented variant; a list nobody can read is refused
    txt = subprocess.run([QVZ_BIN, "-c", "3", inp, out], env=dict(env, QVZ_SWEEP="r:.10,1.00,1.90"), capture_output=True, text=True, check=True).stdout
    got = [ln.split(",")[1].strip() for ln in txt.splitlines() if ln.startswith("rate,")]
    c3 = {p["comp"]: p for p in G.SWEEPS["rd_c3r_t2k"]["points"]}
    assert got == [c3[".10"]["rate"], c3["1.00"]["rate"], c3["1.90"]["rate"]]
    bad = subprocess.run([QVZ_BIN, inp, out], env=dict(env, QVZ_SWEEP="f:0.1;0.2"), capture_output=True, text=True)
    assert bad.returncode == 1 and "QVZ_SWEEP" in bad.stdout


@pytest.mark.parametrize("name,shards", [("rd_t2k", 5), ("rd_c3r_t2k", 3)])
def test_sharded_sweep_decodes_to_the_reference_text(tmp_path, name, shards):
    """The same sweep with the GPU coder (a shard set per point): every set decodes - with the ORACLE's decoder, shard by shard -
    to the -u text the reference wrote at that point, its -u file is that text, and it is the file a separate sharded encode of
    the point writes."""
    from qvz_amd import host
    e = G.SWEEPS[name]
    inp, rows = write_sweep_input(tmp_path, name)
    k, mode, ratios = G.sweep_opts(name)
    hmode = host.MODE_RATIO if mode == O.MODE_RATIO else host.MODE_FIXED
    pick = [0, 1, 7, 10, 19]
    out, u = str(tmp_path / "rd.qset"), str(tmp_path / "rd.u")
    fresh_process_rand()
    res = host.encode_file_sweep(inp, out, [(hmode, ratios[i]) for i in pick], shards=shards, clusters=k, dist="M", u_path=u,
                                 seed_mode=host.SEED_DEBUG)
    for j, i in enumerate(pick):
        pt = e["points"][i]
        ub = open(f"{u}.{j:02d}", "rb").read()
        assert G.md5(ub) == pt["u_md5"], f"point {pt['comp']}: -u differs from the reference's"
        assert f"{res[j].distortion:.4f}" == pt["distortion"]
        prefix = str(tmp_path / f"split{j}")
        host.shardset_split(f"{out}.{j:02d}", prefix)
        dec = b"".join(O.decode(open(f"{prefix}.{g:04d}", "rb").read()) for g in range(shards))
        assert dec == ub, f"point {pt['comp']}: the shard set does not decode to the reference's text"
        alone = str(tmp_path / f"alone{j}.qset")
        fresh_process_rand()
        host.encode_file_shardset(inp, alone, shards, clusters=k, mode=hmode, ratio=ratios[i], dist="M", seed_mode=host.SEED_DEBUG)
        assert open(alone, "rb").read() == open(f"{out}.{j:02d}", "rb").read()


def test_sweep_over_distortion_measures(tmp_path):
    """Points may differ in -d too (the counters do not depend on it): -d L -f 0.5 and -d M -r 1.0 of the t2k input in one job are
    the two stored reference fixtures."""
    from qvz_amd import host
    rows = G.case_rows("t2k_f05_L")
    inp = tmp_path / "t2k.txt"
    inp.write_bytes(rows.tobytes())
    out = str(tmp_path / "m.q")
    fresh_process_rand()
    host.encode_file_sweep(str(inp), out, [(host.MODE_RATIO, 0.5, "L"), (host.MODE_FIXED, 1.0, "M")], seed_mode=host.SEED_DEBUG)
    assert open(out + ".00", "rb").read() == G.gz("t2k_f05_L.q.gz")
    assert open(out + ".01", "rb").read() == G.gz("t2k_r10_M.q.gz")

#!/usr/bin/env python
"""One command that pins this repository to PaddlePaddle itself (VERDICT r04 "next" #1).

Everything under tests/golden/ was produced by the reference's own Python source executed over oracle/paddle_shim, a
torch-backed stand-in whose *kernel* semantics (Linear [in, out], round-half-away, padding_idx, LSTM gate order,
padding="same" with a dilation, the paddle.save layout) are documentation-derived: Paddle cannot be installed in the
build image.  On a machine that has ``paddlepaddle >= 2.1.2`` and a checkout of the reference:

    PARAKEET_REAL_PADDLE=1 PARAKEET_REFERENCE=/path/to/Parakeet python tools/verify_with_paddle.py \\
        --fs2-ckpt  fastspeech2_nosil_ljspeech_ckpt_0.5 \\
        --pwg-ckpt  pwg_ljspeech_ckpt_0.5 \\
        [--waveflow-ckpt waveflow_ljspeech_ckpt_0.3/step-2000000 --waveflow-config waveflow_ljspeech_ckpt_0.3/config.yaml]
    python -m pytest tests -q -m "not gpu"      # oracle  <-> tests/golden_paddle/   (on the same machine)
    python -m pytest tests -q -m gpu            # engine  <-> tests/golden_paddle/   (on an MI355X; the directory travels)

does, in this order:

(a) re-runs every generator (tools/make_golden.py, make_golden_speedyspeech.py, make_golden_ar.py) with REAL Paddle
    executing the reference source, into tests/golden_paddle/ (same file names and keys as tests/golden/);
(b) writes the archive fixtures with Paddle's own ``paddle.save`` (``paddle_written_*``): the first bytes Paddle itself
    wrote that parakeet_amd/checkpoint.py gets to read;
(c) loads released checkpoints with ``paddle.load`` and runs the loop of examples/fastspeech2/ljspeech/synthesize_e2e.py:53-102
    (FastSpeech2 -> ZScore -> Parallel WaveGAN; the frontend replaced by three fixed phone sequences, the vocoder's in-call
    ``paddle.randn`` by a recorded draw) and, if given, examples/waveflow/synthesize.py:31-41; stores ids + mel + noise + wav
    (``released_*.npz``) and copies the checkpoint directories next to them so the tests are self-contained.  Without
    ``--fs2-ckpt/--pwg-ckpt`` the same leg runs on stand-in checkpoint directories of the released layout holding the
    synthetic LJSpeech-configuration weights, written by the backend's own ``paddle.save``;
(d) prints, per tensor, stand-in golden vs Paddle golden (shape, max |diff|) and writes ``report.json``; exit status 1 if
    anything differs by more than ``--tol``.

Without PARAKEET_REAL_PADDLE the very same code runs over the stand-in (how tests/test_verify_paddle_cpu.py exercises it in the
build image: the diff of step (d) is then exactly zero).  The consuming tests are tests/test_golden_*.py etc. (parametrised
over tests/golden_paddle/ by tests/conftest.py when the directory exists) and tests/test_released_ckpt_{cpu,gpu}.py."""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ROOT = ref_import.ROOT
GENERATORS = ["make_golden.py", "make_golden_speedyspeech.py", "make_golden_ar.py"]

# CMU phones with stress marks as g2p_en emits them + the recipe's extra symbols (examples/fastspeech2/ljspeech: the
# released phone_id_map.txt is data-derived, parakeet/datasets/preprocess_utils.py:92-103: <pad>, <unk>, phones, punctuation, <eos>)
_VOWELS = "AA AE AH AO AW AY EH ER EY IH IY OW OY UH UW".split()
_CONS = "B CH D DH F G HH JH K L M N NG P R S SH T TH V W Y Z ZH".split()
STANDIN_PHONES = ["<pad>", "<unk>"] + sorted([v + s for v in _VOWELS for s in "012"] + _CONS + ["sp", "spn"]) + [",", ".", "?", "!", "<eos>"]

# three fixed sentences as phones (what English().phoneticize()[1:-1] gives for them with g2p_en; blanks already dropped)
SENTENCES = OrderedDict([
    ("001", "DH AH0 K W IH1 K B R AW1 N F AA1 K S JH AH1 M P S OW1 V ER0 DH AH0 L EY1 Z IY0 D AO1 G .".split()),
    ("002", "P R IH1 N T IH0 NG , IH0 N DH AH0 OW1 N L IY0 S EH1 N S W IH1 DH W IH1 CH W IY1 AA1 R AE1 T P R EH1 Z AH0 N T K AH0 N S ER1 N D .".split()),
    ("003", "HH AW1 M AH1 CH W UH1 D ? QQ ! Y EH1 S".split()),      # "QQ": not in any map -> "sp" (synthesize_e2e.py:93-96)
])
PUNC = "：，；。？！“”‘’':,;.?!"


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


# ---- (a) generators -------------------------------------------------------------------------------------------------

def run_generators(out, only):
    env = dict(os.environ, PARAKEET_GOLDEN_DIR=out)
    for g in GENERATORS:
        if only and g not in only:
            continue
        print(f"[a] {g} ({ref_import.backend()})", flush=True)
        subprocess.run([sys.executable, os.path.join(HERE, g)], check=True, env=env, cwd=ROOT)


# ---- (b) archives written by the backend's paddle.save ---------------------------------------------------------------

def backend_save(obj, path):
    """``paddle.save(obj, path)`` of the backend: Paddle itself, or tools/make_paddle_fixture.py's restatement of it."""
    def conv(o, leaf):
        if isinstance(o, np.ndarray):
            return leaf(o)
        if isinstance(o, dict):
            return type(o)((k, conv(v, leaf)) for k, v in o.items())
        return o
    if ref_import.REAL:
        import paddle
        paddle.save(conv(obj, paddle.to_tensor), path)
    else:
        import make_paddle_fixture as mpf
        n = [0]

        def leaf(a):
            n[0] += 1
            return mpf.VarBase("param_%d" % n[0], a)
        mpf.paddle_save(conv(obj, leaf), path)


def write_archives(out):
    import make_paddle_fixture as mpf
    t = mpf.tensors(2021)
    archive = {"epoch": 1, "iteration": 7, "main_params": OrderedDict(t),
               "main_optimizer": {"param_0_moment1_0": t["encoder.embed.0.weight"] * np.float32(0.1),
                                  "LR_Scheduler": {"last_epoch": 7, "last_lr": 0.001}}}
    backend_save(archive, os.path.join(out, "paddle_written_updater.pdz"))      # StandardUpdater.state_dict layout
    backend_save(OrderedDict(t), os.path.join(out, "paddle_written_state.pdparams"))   # bare layer.state_dict()
    np.savez(os.path.join(out, "paddle_written_expected.npz"), **t)
    print("[b] paddle_written_updater.pdz, paddle_written_state.pdparams")


# ---- (c) released checkpoints -----------------------------------------------------------------------------------------

def make_standin_checkpoints(out, quick):
    """Directories laid out like fastspeech2_nosil_ljspeech_ckpt_0.5 / pwg_ljspeech_ckpt_0.5 / waveflow_ljspeech_ckpt_0.3
    (examples/fastspeech2/ljspeech/README.md:81-105) with synthetic weights, saved by the backend's paddle.save."""
    import yaml

    from parakeet_amd import synthetic as syn
    base = os.path.join(out, "released")
    fdir, pdir, wdir = (os.path.join(base, d) for d in ("fastspeech2_standin_ckpt", "pwg_standin_ckpt", "waveflow_standin_ckpt"))
    for d in (fdir, pdir, wdir):
        os.makedirs(d, exist_ok=True)
    fcfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "fixtures", "fastspeech2_ljspeech.yaml")))
    pcfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "fixtures", "pwg_ljspeech.yaml")))
    if quick:
        fcfg["model"].update(elayers=1, dlayers=1)
        pcfg["generator_params"].update(layers=6, stacks=3)
    idim = len(STANDIN_PHONES)
    fstate = syn.fastspeech2_state(idim, 80, dict(syn.FS2_LJSPEECH, **{k: fcfg["model"][k] for k in ("elayers", "dlayers")}), seed=31)
    gp = pcfg["generator_params"]
    pstate = syn.pwg_state(dict(syn.PWG_LJSPEECH, layers=gp["layers"], stacks=gp["stacks"]), seed=32, weight_norm=True)
    yaml.safe_dump(fcfg, open(os.path.join(fdir, "default.yaml"), "wt"))
    yaml.safe_dump(pcfg, open(os.path.join(pdir, "pwg_default.yaml"), "wt"))
    with open(os.path.join(fdir, "phone_id_map.txt"), "wt") as f:
        for i, p in enumerate(STANDIN_PHONES):
            f.write(f"{p} {i}\n")
    np.save(os.path.join(fdir, "speech_stats.npy"), np.stack(syn.mel_stats(seed=7)).astype(np.float32))
    np.save(os.path.join(pdir, "pwg_stats.npy"), np.stack(syn.mel_stats(seed=8)).astype(np.float32))
    opt = {"LR_Scheduler": {"last_epoch": 3, "last_lr": 0.001}}
    backend_save({"epoch": 1, "iteration": 100000, "main_params": OrderedDict(fstate), "main_optimizer": opt},
                 os.path.join(fdir, "snapshot_iter_100000.pdz"))
    backend_save({"epoch": 1, "iteration": 400000, "generator_params": OrderedDict(pstate), "generator_optimizer": opt},
                 os.path.join(pdir, "pwg_snapshot_iter_400000.pdz"))
    wcfg = dict(syn.WAVEFLOW_LJSPEECH, channels=64)
    if quick:
        wcfg.update(n_flows=2)       # n_layers is tied to the dilation table (waveflow.py:330)
    wstate = syn.waveflow_state(wcfg, seed=33, weight_norm=True)
    model = {k: wcfg[k] for k in ("upsample_factors", "n_flows", "n_layers", "n_group", "channels", "kernel_size")}
    yaml.safe_dump({"data": {"n_mels": 80}, "model": model}, open(os.path.join(wdir, "config.yaml"), "wt"))
    backend_save(OrderedDict(wstate), os.path.join(wdir, "step-2000000.pdparams"))
    return fdir, pdir, os.path.join(wdir, "step-2000000"), os.path.join(wdir, "config.yaml")


def _one(d, pattern):
    import glob
    hits = sorted(glob.glob(os.path.join(d, pattern)))
    if len(hits) != 1:
        raise SystemExit(f"{d}: expected exactly one {pattern}, found {hits}")
    return hits[0]


def released_e2e(out, tag, fdir, pdir):
    """examples/fastspeech2/ljspeech/synthesize_e2e.py:45-102 with the backend's Paddle; the frontend is SENTENCES."""
    import paddle
    import yaml
    fsm = ref_import.load("parakeet.models.fastspeech2.fastspeech2")
    pw = ref_import.load("parakeet.models.parallel_wavegan.parallel_wavegan")
    norm = ref_import.load("parakeet.modules.normalizer")
    fcfg = yaml.safe_load(open(_one(fdir, "*default.yaml")))
    pcfg = yaml.safe_load(open(_one(pdir, "*default.yaml")))
    with open(os.path.join(fdir, "phone_id_map.txt")) as f:                      # :45-50
        phn_id = [line.strip().split() for line in f.readlines()]
    vocab_size = len(phn_id)
    phone_id_map = {phn: int(i) for phn, i in phn_id}
    model = fsm.FastSpeech2(idim=vocab_size, odim=fcfg["n_mels"], **fcfg["model"])   # :53-54
    fckpt, pckpt = _one(fdir, "snapshot_iter_*.pdz"), _one(pdir, "*snapshot_iter_*.pdz")
    model.set_state_dict(paddle.load(fckpt)["main_params"])                      # :56-57
    model.eval()
    vocoder = pw.PWGGenerator(**pcfg["generator_params"])                        # :60
    vocoder.set_state_dict(paddle.load(pckpt)["generator_params"])
    vocoder.remove_weight_norm()
    vocoder.eval()
    mu, std = np.load(_one(fdir, "*stats.npy"))                                  # :70-80
    fnorm = norm.ZScore(paddle.to_tensor(mu), paddle.to_tensor(std))
    mu, std = np.load(_one(pdir, "*stats.npy"))
    pnorm = norm.ZScore(paddle.to_tensor(mu), paddle.to_tensor(std))
    fs2_inference = fsm.FastSpeech2Inference(fnorm, model)                       # :82-83
    pwg_inference = pw.PWGInference(pnorm, vocoder)
    rng = np.random.default_rng(20210815)
    rec = {"fs2_dir": np.array(os.path.basename(fdir)), "pwg_dir": np.array(os.path.basename(pdir)),
           "hop": np.array(int(np.prod(pcfg["generator_params"]["upsample_scales"])))}
    for utt, phones in SENTENCES.items():                                        # :88-102
        phones = [p if (p in phone_id_map and p not in PUNC) else "sp" for p in phones]
        ids = np.array([phone_id_map[p] for p in phones], np.int64)
        with paddle.no_grad():
            mel = fs2_inference(paddle.to_tensor(ids))
            noise = rng.standard_normal(int(mel.shape[0]) * int(rec["hop"])).astype(np.float32)
            with ref_import.fixed_randn(noise):
                wav = pwg_inference(mel)
        rec[f"ids_{utt}"], rec[f"mel_{utt}"] = ids, mel.numpy().astype(np.float32)
        rec[f"noise_{utt}"], rec[f"wav_{utt}"] = noise, wav.numpy().astype(np.float32).reshape(-1)
        print(f"[c] {tag} {utt}: {len(ids)} phones -> {rec[f'mel_{utt}'].shape[0]} frames -> {rec[f'wav_{utt}'].size} samples")
    np.savez_compressed(os.path.join(out, f"released_{tag}.npz"), **rec)
    return {"fs2": {os.path.basename(p): sha256(p) for p in (fckpt,)}, "pwg": {os.path.basename(p): sha256(p) for p in (pckpt,)}}


def released_waveflow(out, tag, ckpt, config):
    """examples/waveflow/synthesize.py:31-41 without the AMP context (fp32 reference; the engine's "f16" mode has its own bar)."""
    import paddle
    import yaml
    wfm = ref_import.load("parakeet.models.waveflow")
    cfg = yaml.safe_load(open(config))
    m = cfg["model"]
    model = wfm.ConditionalWaveFlow(upsample_factors=m["upsample_factors"], n_flows=m["n_flows"], n_layers=m["n_layers"],
                                    n_group=m["n_group"], channels=m["channels"], n_mels=cfg["data"]["n_mels"],
                                    kernel_size=m["kernel_size"])
    path = ckpt if ckpt.endswith(".pdparams") else ckpt + ".pdparams"
    model.set_state_dict(paddle.load(path))                                      # utils/checkpoint.py:61-108
    for layer in model.sublayers():                                              # layer_tools.py:40-46
        try:
            paddle.nn.utils.remove_weight_norm(layer)
        except ValueError:
            pass
    model.eval()
    rng = np.random.default_rng(20210816)
    mel = np.maximum(rng.normal(-4, 2, size=(1, 80, 12)), np.log(1e-5)).astype(np.float32)   # natural-log mel domain
    t = mel.shape[-1]
    for f in m["upsample_factors"]:
        t = f * t - f                                                           # trim_conv_artifact (waveflow.py:103-132)
    z = rng.standard_normal((1, t)).astype(np.float32)
    with ref_import.fixed_randn(z), paddle.no_grad():
        wav = model.infer(paddle.to_tensor(mel)).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(out, f"released_waveflow_{tag}.npz"), mel=mel, z=z, wav=wav,
                        ckpt=np.array(os.path.basename(path)), config=np.array(os.path.basename(config)),
                        dir=np.array(os.path.basename(os.path.dirname(os.path.abspath(path)))))
    print(f"[c] waveflow {tag}: mel {mel.shape} -> wav {wav.shape}")
    return {os.path.basename(path): sha256(path)}


def copy_dir(src, out):
    dst = os.path.join(out, "released", os.path.basename(os.path.normpath(src)))
    if os.path.abspath(src) != os.path.abspath(dst):
        shutil.copytree(src, dst, dirs_exist_ok=True)
    return dst


# ---- (d) stand-in goldens vs this run ---------------------------------------------------------------------------------

def diff_report(out, tol):
    base = os.path.join(ROOT, "tests", "golden")
    rows, worst = [], 0.0
    for name in sorted(os.listdir(out)):
        if not name.endswith(".npz") or not os.path.exists(os.path.join(base, name)):
            continue
        a, b = np.load(os.path.join(base, name), allow_pickle=False), np.load(os.path.join(out, name), allow_pickle=False)
        for k in sorted(set(a.files) | set(b.files)):
            if k not in a.files or k not in b.files:
                rows.append((name, k, "only in " + ("stand-in" if k in a.files else ref_import.backend()), None))
                continue
            x, y = a[k], b[k]
            if x.shape != y.shape:
                rows.append((name, k, f"shape {x.shape} vs {y.shape}", float("inf")))
                worst = float("inf")
            elif x.dtype.kind in "fc":
                d = float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) if x.size else 0.0
                rows.append((name, k, "", d))
                worst = max(worst, d)
            else:
                d = 0.0 if np.array_equal(x, y) else float("inf")
                rows.append((name, k, "" if d == 0 else "values differ", d))
                worst = max(worst, d)
    print(f"[d] stand-in goldens (tests/golden) vs {ref_import.backend()} goldens ({out}):")
    for name, k, note, d in rows:
        if note or (d is not None and d > tol):
            print(f"    {name:36s} {k:28s} {note} {'' if d is None else '%.3g' % d}")
    finite = [d for *_, d in rows if d is not None]
    print(f"    {len(rows)} tensors compared, {sum(1 for d in finite if d > tol)} above tol={tol:g}, worst {worst:.3g}")
    return rows, worst


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=None, help="default: tests/golden_paddle (real Paddle) / required for the stand-in")
    ap.add_argument("--fs2-ckpt", help="unzipped fastspeech2_nosil_ljspeech_ckpt_0.5")
    ap.add_argument("--pwg-ckpt", help="unzipped pwg_ljspeech_ckpt_0.5")
    ap.add_argument("--waveflow-ckpt", help="e.g. waveflow_ljspeech_ckpt_0.3/step-2000000")
    ap.add_argument("--waveflow-config", help="the yaml next to it (model + data sections)")
    ap.add_argument("--only", default="", help="comma-separated subset of generators, e.g. make_golden_speedyspeech.py")
    ap.add_argument("--skip-generators", action="store_true")
    ap.add_argument("--quick", action="store_true", help="reduced depth for the stand-in checkpoint directories")
    ap.add_argument("--tol", type=float, default=1e-4)
    a = ap.parse_args()
    if a.out is None:
        if not ref_import.REAL:
            raise SystemExit("stand-in run: give --out (tests/golden_paddle is reserved for files real Paddle produced)")
        a.out = os.path.join(ROOT, "tests", "golden_paddle")
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    os.environ["PARAKEET_GOLDEN_DIR"] = out
    ref_import.setup()
    import paddle
    version = getattr(paddle, "__version__", "stand-in (oracle/paddle_shim)")
    print(f"backend: {ref_import.backend()}  paddle {version}  reference {ref_import.REF}  ->  {out}")
    if not a.skip_generators:
        run_generators(out, [g for g in a.only.split(",") if g])
    write_archives(out)
    manifest = {"backend": ref_import.backend(), "paddle_version": version, "released": {}}
    sf, sp, sw, swc = make_standin_checkpoints(out, a.quick)
    pairs = [("standin", sf, sp)]
    if a.fs2_ckpt and a.pwg_ckpt:
        pairs.append(("ljspeech", copy_dir(a.fs2_ckpt, out), copy_dir(a.pwg_ckpt, out)))
    for tag, fdir, pdir in pairs:
        manifest["released"][tag] = released_e2e(out, tag, fdir, pdir)
    flows = [("standin", sw, swc)]
    if a.waveflow_ckpt:
        wd = copy_dir(os.path.dirname(os.path.abspath(a.waveflow_ckpt)), out)
        cfgp = a.waveflow_config or _one(wd, "*.yaml")
        if os.path.dirname(os.path.abspath(cfgp)) != wd:
            shutil.copy(cfgp, wd)
        flows.append(("ljspeech", os.path.join(wd, os.path.basename(a.waveflow_ckpt)), os.path.join(wd, os.path.basename(cfgp))))
    for tag, ckpt, cfgp in flows:
        manifest["released"]["waveflow_" + tag] = released_waveflow(out, tag, ckpt, cfgp)
    rows, worst = diff_report(out, a.tol)
    manifest["diff"] = {"tol": a.tol, "worst": None if worst == float("inf") else worst, "infinite": worst == float("inf"),
                        "rows": [[n, k, note, None if d is None or d == float("inf") else d] for n, k, note, d in rows]}
    json.dump(manifest, open(os.path.join(out, "report.json"), "wt"), indent=1)
    print("wrote", os.path.join(out, "report.json"))
    return 0 if worst <= a.tol else 1


if __name__ == "__main__":
    sys.exit(main())

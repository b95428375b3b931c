"""Checks shared by the tests that consume what tools/verify_with_paddle.py writes: ``released_*.npz`` (ids / mel / noise /
wav of the loop of examples/fastspeech2/ljspeech/synthesize_e2e.py:53-102 run by the reference's source on checkpoint
directories of the released layout) and ``paddle_written_*`` (archives saved by the backend's own ``paddle.save``).

``base`` holds the npz files, ``base/released/<dir>`` the checkpoint directories they name."""
import glob
import os

import numpy as np
import torch

from parakeet_amd import checkpoint as ck

UTTS = ("001", "002", "003")


def _one(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, pattern)))
    assert len(hits) == 1, (d, pattern, hits)
    return hits[0]


def released_files(base):
    return sorted(p for p in glob.glob(os.path.join(base, "released_*.npz")) if "waveflow" not in os.path.basename(p))


def waveflow_files(base):
    return sorted(glob.glob(os.path.join(base, "released_waveflow_*.npz")))


def _dirs(npz, ckpt_root):
    g = np.load(npz)
    return g, os.path.join(ckpt_root, str(g["fs2_dir"])), os.path.join(ckpt_root, str(g["pwg_dir"]))


def check_oracle_released(npz, ckpt_root, mel_tol=2e-5, wav_tol=2e-5):
    """oracle (torch CPU restatement) <-> the reference's source, on weights READ FROM THE CHECKPOINT FILES by
    parakeet_amd.checkpoint (so the reader, the key names, the weight-norm fold and the stats files are all in the loop)."""
    import yaml

    from oracle import fastspeech2_ref, pwg_ref
    g, fdir, pdir = _dirs(npz, ckpt_root)
    fcfg = yaml.safe_load(open(_one(fdir, "*default.yaml")))
    pcfg = yaml.safe_load(open(_one(pdir, "*default.yaml")))
    fstate = ck.load_params(_one(fdir, "snapshot_iter_*.pdz"), "main_params")
    pstate = ck.load_params(_one(pdir, "*snapshot_iter_*.pdz"), "generator_params")
    fmu, fsd = ck.load_stats(_one(fdir, "*stats.npy"))
    pmu, psd = ck.load_stats(_one(pdir, "*stats.npy"))
    fo = {k: fcfg["model"][k] for k in fastspeech2_ref.DEFAULT_CFG if k in fcfg["model"]}
    gp = pcfg["generator_params"]
    po = {k: gp[k] for k in ("layers", "stacks", "kernel_size", "aux_context_window", "upsample_scales")}
    worst = {"mel": 0.0, "wav": 0.0}
    for u in UTTS:
        mel = fastspeech2_ref.fastspeech2_inference(fstate, fmu, fsd, g[f"ids_{u}"], fo).numpy()
        assert mel.shape == g[f"mel_{u}"].shape, (u, mel.shape, g[f"mel_{u}"].shape)       # integer durations equal
        worst["mel"] = max(worst["mel"], float(np.abs(mel - g[f"mel_{u}"]).max()))
        wav = pwg_ref.pwg_inference(pstate, pmu, psd, torch.from_numpy(g[f"mel_{u}"]), torch.from_numpy(g[f"noise_{u}"]), po)
        wav = wav.numpy().reshape(-1)
        assert wav.shape == g[f"wav_{u}"].shape
        worst["wav"] = max(worst["wav"], float(np.abs(wav - g[f"wav_{u}"]).max()))
    assert worst["mel"] < mel_tol and worst["wav"] < wav_tol, worst
    return worst


def check_engine_released(npz, ckpt_root):
    """HIP engine, built from the checkpoint directories the way the recipe builds the reference models, <-> the
    reference's source: durations equal, mel L1 < 1e-4 (north_star) and < 1e-5 (10x the measured error), wav < 1e-4 rel."""
    g, fdir, pdir = _dirs(npz, ckpt_root)
    am, table = ck.load_fastspeech2(_one(fdir, "*default.yaml"), _one(fdir, "snapshot_iter_*.pdz"),
                                    _one(fdir, "*stats.npy"), os.path.join(fdir, "phone_id_map.txt"))
    voc = ck.load_pwg(_one(pdir, "*default.yaml"), _one(pdir, "*snapshot_iter_*.pdz"), _one(pdir, "*stats.npy"))
    out = {}
    for u in UTTS:
        mel = am(g[f"ids_{u}"]).numpy()
        assert mel.shape == g[f"mel_{u}"].shape, (u, mel.shape, g[f"mel_{u}"].shape)
        l1 = float(np.abs(mel - g[f"mel_{u}"]).mean())
        assert l1 < 1e-4 and l1 < 1e-5, (u, l1)
        wav = voc(g[f"mel_{u}"], noise=g[f"noise_{u}"]).numpy().reshape(-1)
        rel = float(np.abs(wav - g[f"wav_{u}"]).max() / np.abs(g[f"wav_{u}"]).max())
        assert rel < 1e-4, (u, rel)
        out[u] = (l1, rel)
    return out


def _waveflow(npz, ckpt_root):
    import yaml
    g = np.load(npz)
    d = os.path.join(ckpt_root, str(g["dir"]))
    return g, yaml.safe_load(open(os.path.join(d, str(g["config"])))), os.path.join(d, str(g["ckpt"]))


def check_oracle_waveflow(npz, ckpt_root):
    from oracle import waveflow_ref
    g, cfg, path = _waveflow(npz, ckpt_root)
    m = dict(cfg["model"], n_mels=cfg["data"]["n_mels"])
    wav = waveflow_ref.infer(ck.load_params(path), torch.from_numpy(g["mel"]), torch.from_numpy(g["z"]), m).numpy()
    assert wav.shape == g["wav"].shape
    err = float(np.abs(wav - g["wav"]).max() / max(1.0, np.abs(g["wav"]).max()))
    assert err < 1e-5, err
    return err


def check_engine_waveflow(npz, ckpt_root):
    g, cfg, path = _waveflow(npz, ckpt_root)
    model = ck.load_waveflow(cfg, path)
    wav = model.infer(g["mel"], z=g["z"]).numpy()
    assert wav.shape == g["wav"].shape
    err = float(np.abs(wav - g["wav"]).max() / max(1.0, np.abs(g["wav"]).max()))
    assert err < 1e-4, err
    return err


def check_paddle_written(base):
    """Archives saved by the backend's own paddle.save (``paddle_written_*``) read back through parakeet_amd.checkpoint."""
    with np.load(os.path.join(base, "paddle_written_expected.npz")) as z:
        want = {k: z[k] for k in z.files}
    got = ck.load_params(os.path.join(base, "paddle_written_updater.pdz"), "main_params")
    assert list(got) == list(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k])
    arch = ck.load_archive(os.path.join(base, "paddle_written_updater.pdz"))
    assert arch["epoch"] == 1 and arch["iteration"] == 7 and arch["main_optimizer"]["LR_Scheduler"]["last_epoch"] == 7
    np.testing.assert_array_equal(arch["main_optimizer"]["param_0_moment1_0"], want["encoder.embed.0.weight"] * np.float32(0.1))
    got = ck.load_params(os.path.join(base, "paddle_written_state.pdparams"))
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k])

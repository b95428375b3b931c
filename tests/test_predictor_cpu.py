"""The paddle.inference-shaped wrapper (host logic only; the GPU path is exercised in test_speedyspeech_gpu)."""
import os

import numpy as np
import pytest

from parakeet_amd.predictor import Config, create_predictor


def test_predictor_protocol():
    calls = []

    def model(phones, tones):
        calls.append((phones.copy(), tones.copy()))
        return (phones + tones).astype(np.float32)[:, None] * np.ones((1, 3), np.float32)

    p = create_predictor(model, ["phones", "tones"])
    assert p.get_input_names() == ["phones", "tones"] and p.get_output_names() == ["out"]
    with pytest.raises(RuntimeError):
        p.run()
    a, b = np.arange(6, dtype=np.int64), np.ones(6, dtype=np.int64)
    for name, v in zip(p.get_input_names(), (a, b)):
        h = p.get_input_handle(name)
        h.reshape(v.shape)
        h.copy_from_cpu(v)
    with pytest.raises(RuntimeError):
        p.get_output_handle("out").copy_to_cpu()
    assert p.run()
    out = p.get_output_handle(p.get_output_names()[0]).copy_to_cpu()
    assert out.shape == (6, 3) and np.array_equal(out[:, 0], (a + b).astype(np.float32))
    assert len(calls) == 1


def test_config_resolves_the_recipe_artefacts(tmp_path):
    """Config takes the reference script's two paths (examples/speedyspeech/baker/inference.py:53-66) and finds the recipe's
    own files next to them; several snapshots -> the last; explicit arguments win; unknown kinds and missing files fail loudly;
    there is no CPU execution path to enable."""
    d = tmp_path
    for name in ("speedyspeech.yaml", "snapshot_iter_100.pdz", "snapshot_iter_76000.pdz", "speech_stats.npy", "phone_id_map.txt",
                 "tone_id_map.txt", "pwg_default.yaml", "pwg_snapshot_iter_400000.pdz", "pwg_stats.npy"):
        (d / name).write_bytes(b"")
    c = Config(str(d / "speedyspeech.pdmodel"), str(d / "speedyspeech.pdiparams"))
    c.enable_use_gpu(100, 0)
    c.enable_memory_optim()
    a = c.resolve()
    assert c.model == "speedyspeech" and c.device_id == 0 and c.memory_optim
    assert [os.path.basename(a[k]) for k in ("config", "checkpoint", "stat", "phones_dict", "tones_dict")] == [
        "speedyspeech.yaml", "snapshot_iter_76000.pdz", "speech_stats.npy", "phone_id_map.txt", "tone_id_map.txt"]
    p = Config(str(d / "pwg.pdmodel"), str(d / "pwg.pdiparams")).resolve()
    assert [os.path.basename(p[k]) for k in ("config", "checkpoint", "stat")] == [
        "pwg_default.yaml", "pwg_snapshot_iter_400000.pdz", "pwg_stats.npy"] and p["phones_dict"] is None
    e = Config(model="fastspeech2", model_dir=str(d), config="x.yaml", checkpoint="y.pdz", stat="z.npy", phones_dict="m.txt").resolve()
    assert (e["config"], e["checkpoint"], e["stat"], e["phones_dict"]) == ("x.yaml", "y.pdz", "z.npy", "m.txt")
    with pytest.raises(ValueError):
        Config(str(d / "tacotron9.pdmodel"))
    with pytest.raises(ValueError):
        Config()
    (d / "sub").mkdir()
    with pytest.raises(FileNotFoundError):
        Config(model="pwg", model_dir=str(d / "sub")).resolve()
    with pytest.raises(RuntimeError):
        c.disable_gpu()


def test_nets_utils_worked_examples():
    """The docstring examples of parakeet/modules/nets_utils.py:36-42,71-75,119-123 and fastspeech2.py:634-637."""
    from parakeet_amd import nets_utils as nu
    x = [np.ones(4), np.ones(2), np.ones(1)]
    assert np.array_equal(nu.pad_list(x, 0), [[1, 1, 1, 1], [1, 1, 0, 0], [1, 0, 0, 0]])
    assert np.array_equal(nu.make_pad_mask([5, 3, 2]).astype(int), [[0, 0, 0, 0, 0], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1]])
    assert np.array_equal(nu.make_non_pad_mask(np.array([5, 3, 2])).astype(int),
                          [[1, 1, 1, 1, 1], [1, 1, 1, 0, 0], [1, 1, 0, 0, 0]])
    assert nu.source_mask([5, 3]).shape == (2, 1, 5)
    with pytest.raises(ValueError):
        nu.make_pad_mask([3], length_dim=0)


def test_snapshot_choice_is_numeric_and_never_crosses_model_kinds(tmp_path):
    """"snapshot_iter_9999" sorts after "snapshot_iter_10000" as a string; and in a directory that holds two
    model kinds the catch-all patterns must not hand one model the other's files."""
    from parakeet_amd.predictor import Config
    d = tmp_path / "inference"
    d.mkdir()
    for name in ("speedyspeech.yaml", "snapshot_iter_9999.pdz", "snapshot_iter_10000.pdz", "speech_stats.npy", "phone_id_map.txt",
                 "tone_id_map.txt"):
        (d / name).write_bytes(b"")
    a = Config(str(d / "speedyspeech.pdmodel")).resolve()
    assert os.path.basename(a["checkpoint"]) == "snapshot_iter_10000.pdz"
    # single-kind directory: the fall-backs still work for the vocoder (a bare pwg directory with odd names)
    v = tmp_path / "voc"
    v.mkdir()
    for name in ("default.yaml", "gen.pdz", "feats_stats.npy"):
        (v / name).write_bytes(b"")
    p = Config(model="pwg", model_dir=str(v)).resolve()
    assert [os.path.basename(p[k]) for k in ("config", "checkpoint", "stat")] == ["default.yaml", "gen.pdz", "feats_stats.npy"]
    # shared directory without the vocoder's own files: refuse instead of taking the acoustic model's
    (d / "pwg.pdmodel").write_bytes(b"")
    import pytest
    with pytest.raises(FileNotFoundError, match="several model kinds"):
        Config(str(d / "pwg.pdmodel")).resolve()
    (d / "pwg_default.yaml").write_bytes(b"")
    (d / "pwg_snapshot_iter_400000.pdz").write_bytes(b"")
    with pytest.raises(FileNotFoundError, match="stat="):
        Config(str(d / "pwg.pdmodel")).resolve()
    (d / "pwg_stats.npy").write_bytes(b"")
    p = Config(str(d / "pwg.pdmodel")).resolve()
    assert os.path.basename(p["checkpoint"]) == "pwg_snapshot_iter_400000.pdz" and os.path.basename(p["stat"]) == "pwg_stats.npy"
    # and the acoustic model of that directory never sees the pwg_* files through a wildcard
    (d / "snapshot_iter_9999.pdz").unlink()
    (d / "snapshot_iter_10000.pdz").unlink()
    with pytest.raises(FileNotFoundError):
        Config(str(d / "speedyspeech.pdmodel")).resolve()

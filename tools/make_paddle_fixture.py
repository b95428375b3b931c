#!/usr/bin/env python
"""Write the checkpoint fixtures tests/golden/paddle21_*.{pdz,pdparams}: archives laid out the way
``paddle.save`` of Paddle 2.1.x lays them out (python/paddle/framework/io.py), produced WITHOUT Paddle.

Paddle cannot be installed here, so the two code paths of ``paddle.save`` are restated below around a stand-in
tensor class; everything that determines the bytes -- ``pickle.Pickler`` with a ``dispatch_table``, protocol 2,
numpy's own ndarray reduction -- is the real machinery:

  save(obj, path, protocol=2)
    _is_state_dict(obj)          every value a tensor (or a dict without Paddle types)  -> _legacy_save
        _build_saved_state_dict  tensor -> ndarray; "StructuredToParameterName@@" = {key: tensor.name}
        _unpack_saved_dict       protocol 2/3: arrays above MAX elements -> flat slices "<key>@@.<i>" +
                                 "UnpackBigParamInfor@@" = {key: {"OriginShape", "slices"}}
        pickle.dump(saved_obj, f, protocol=protocol)
    otherwise                    (the updater archive of parakeet/training/updaters/standard_updater.py:183-190)
        _pickle_save             Pickler.dispatch_table[VarBase] = reduce_varbase -> (tuple, ((name, ndarray),))
                                 Pickler.dispatch_table[LoDTensor] = reduce_LoDTensor -> (eval, ('data', {'data': a}))

Run from the repository root:  python tools/make_paddle_fixture.py
"""
import copyreg
import math
import os
import pickle
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


class VarBase:
    """Stand-in for paddle.fluid.core.VarBase: a named tensor with .numpy()."""

    def __init__(self, name, data):
        self.name, self._data = name, np.asarray(data)

    def numpy(self):
        return self._data


class LoDTensor(VarBase):
    pass


def reduce_varbase(self):
    return (tuple, ((self.name, self.numpy()),))


def reduce_lodtensor(self):
    return (eval, ("data", {"data": self.numpy()}))


def _is_state_dict(obj):
    if not isinstance(obj, dict):
        return False
    for value in obj.values():
        if isinstance(value, dict):
            if any(isinstance(v, VarBase) for v in value.values()):
                return False
        elif not isinstance(value, VarBase):
            return False
    return True


def _build_saved_state_dict(state_dict):
    save_dict, name_table = {}, {}
    for key, value in state_dict.items():
        if isinstance(value, VarBase):
            save_dict[key] = value.numpy()
            name_table[key] = value.name
        else:
            save_dict[key] = value
    save_dict["StructuredToParameterName@@"] = name_table
    return save_dict


def _unpack_saved_dict(saved_obj, protocol, max_bytes=2 ** 30 - 1):
    temp, info = {}, {}
    if 1 < protocol < 4 and isinstance(saved_obj, dict):
        for key, value in saved_obj.items():
            if isinstance(value, np.ndarray):
                max_elems = int(max_bytes / value.dtype.itemsize)
                n = int(np.prod(value.shape))
                if n > max_elems:
                    info[key] = {"OriginShape": value.shape, "slices": []}
                    flat = value.flatten()
                    for i in range(int(math.ceil(n * 1.0 / max_elems))):
                        part = key + "@@." + str(i)
                        info[key]["slices"].append(part)
                        temp[part] = flat[i * max_elems:max_elems * (i + 1)]
    if info:
        for key, value in info.items():
            if key in saved_obj:
                saved_obj.pop(key)
                for part in value["slices"]:
                    saved_obj[part] = temp[part]
        saved_obj["UnpackBigParamInfor@@"] = info
    return saved_obj


def paddle_save(obj, path, protocol=2, max_bytes=2 ** 30 - 1):
    if _is_state_dict(obj):
        saved = _unpack_saved_dict(_build_saved_state_dict(obj), protocol, max_bytes)
        with open(path, "wb") as f:
            pickle.dump(saved, f, protocol=protocol)
        return
    with open(path, "wb") as f:
        pickler = pickle.Pickler(f, protocol)
        pickler.dispatch_table = copyreg.dispatch_table.copy()
        pickler.dispatch_table[VarBase] = reduce_varbase
        pickler.dispatch_table[LoDTensor] = reduce_lodtensor
        pickler.dump(obj)


def tensors(seed):
    rng = np.random.default_rng(seed)
    return OrderedDict([
        ("encoder.embed.0.weight", rng.normal(size=(9, 4)).astype(np.float32)),
        ("encoder.encoders.0.self_attn.linear_q.weight", rng.normal(size=(4, 4)).astype(np.float32)),
        ("postnet.postnet.0.1._variance", rng.uniform(0.5, 1.5, size=(5,)).astype(np.float32)),
        ("conv_layers.0.conv.weight_g", rng.normal(size=(6,)).astype(np.float32)),
        ("conv_layers.0.conv.weight_v", rng.normal(size=(6, 2, 3)).astype(np.float32)),
    ])


def main():
    os.makedirs(OUT, exist_ok=True)
    t = tensors(2021)
    params = OrderedDict((k, VarBase("param_%d" % i, v)) for i, (k, v) in enumerate(t.items()))
    # StandardUpdater.state_dict (standard_updater.py:183-190) + UpdaterBase.save (updater.py:77-80)
    moments = {"param_0_moment1_0": VarBase("param_0_moment1_0", t["encoder.embed.0.weight"] * 0.1),
               "param_0_beta1_pow_acc_0": LoDTensor("beta1_pow", np.asarray([0.9 ** 7], np.float32)),
               "LR_Scheduler": {"last_epoch": 7, "last_lr": 0.001}}
    archive = {"epoch": 1, "iteration": 7, "main_params": params, "main_optimizer": moments}
    paddle_save(archive, os.path.join(OUT, "paddle21_updater.pdz"))
    # a bare layer.state_dict() (utils/checkpoint.py:61-108): _legacy_save with the name table ...
    paddle_save(OrderedDict(params), os.path.join(OUT, "paddle21_state.pdparams"))
    # ... and with the big-parameter split forced by a small element limit, to pin the re-merge
    paddle_save(OrderedDict(params), os.path.join(OUT, "paddle21_state_sliced.pdparams"), max_bytes=40)
    np.savez(os.path.join(OUT, "paddle21_expected.npz"), **t)
    for n in sorted(os.listdir(OUT)):
        if n.startswith("paddle21_"):
            print(n, os.path.getsize(os.path.join(OUT, n)), "bytes")


if __name__ == "__main__":
    main()

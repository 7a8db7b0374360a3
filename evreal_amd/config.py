"""JSON config access with the reference's semantics (utils/util.py:9-12, eval.py:28-35,82-89,118-121)."""
import json
import os
from collections import OrderedDict
from pathlib import Path


def read_json(fname):
    with Path(fname).open('rt', encoding="utf-8") as handle:
        return json.load(handle, object_hook=OrderedDict)


def _named(kind, names):
    out = []
    for name in names:
        cfg = read_json(os.path.join("config", kind, name + ".json"))
        cfg['name'] = name
        out.append(cfg)
    return out


def get_eval_configs(eval_config_names):
    return _named("eval", eval_config_names)


def get_dataset_configs(dataset_names):
    return _named("dataset", dataset_names)


def get_method_config(method_name):
    return read_json(os.path.join("config", "method", method_name + ".json"))

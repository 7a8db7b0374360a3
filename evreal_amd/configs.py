"""Writes the reference's config/ tree (config/{eval,method,dataset}/*.json) from compact tables.

The evaluation surface is three JSON namespaces resolved by name relative to the working directory
(eval.py:28-35,82-89,118-121 of the reference).  `evreal_amd.eval` reads exactly those paths, so it runs
against the reference's own config/ directory unchanged; this module regenerates an equivalent tree
(keys and values as enumerated in SURVEY.md sections 2.3 and 5) for use without the reference checkout:

    python -m evreal_amd.configs [target_dir]
"""
import json
import os
import sys


def _eval_cfg(voxel_method, keep_ratio=1.0, save_images=False, color=None, eval_infer_all=False):
    cfg = {"dataset_kwargs": {"num_bins": 5, "voxel_method": voxel_method, "keep_ratio": keep_ratio},
           "save_images": save_images, "histeq": "none"}
    if color is not None:
        cfg["color"] = color
    cfg.update({"eval_infer_all": eval_infer_all, "ts_tol_ms": 1.0, "create_video": False})
    return cfg


def eval_configs():
    out = {"std": _eval_cfg({"method": "between_frames"}, save_images=True),
           "std_all": _eval_cfg({"method": "between_frames"}, save_images=True, eval_infer_all=True),
           "color": _eval_cfg({"method": "between_frames"}, save_images=True, color=True)}
    for k in range(5, 50, 5):
        out[f"k{k}k"] = _eval_cfg({"method": "k_events", "k": k * 1000, "sliding_window_w": 0})
    for t in range(10, 110, 10):
        out[f"t{t}ms"] = _eval_cfg({"method": "t_seconds", "t": t / 1000.0, "sliding_window_t": 0})
    for r in range(1, 11):
        out[f"kr{r / 10:.1f}"] = _eval_cfg({"method": "between_frames"}, keep_ratio=r / 10)
    return out


def method_configs():
    table = {  # name: (event_tensor_normalization, post_process_norm)   (config/method/*.json:4-5)
        "E2VID": (True, "robust"), "FireNet": (True, "none"), "E2VID+": (False, "none"), "FireNet+": (False, "none"),
        "SPADE-E2VID": (False, "none"), "SSL-E2VID": (False, "exprobust"), "ET-Net": (False, "none"),
        "HyperE2VID": (False, "none")}
    return {n: {"model_name": n, "model_path": f"pretrained/{n}/model.pth", "event_tensor_normalization": a,
                "post_process_norm": b} for n, (a, b) in table.items()}


def write(target="config"):
    for sub, cfgs in (("eval", eval_configs()), ("method", method_configs())):
        os.makedirs(os.path.join(target, sub), exist_ok=True)
        for name, cfg in cfgs.items():
            with open(os.path.join(target, sub, name + ".json"), "w") as f:
                json.dump(cfg, f, indent=4)
    os.makedirs(os.path.join(target, "dataset"), exist_ok=True)


if __name__ == "__main__":
    write(sys.argv[1] if len(sys.argv) > 1 else "config")

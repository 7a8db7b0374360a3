#!/usr/bin/env python3
"""Freeze the reference's THIRD-PARTY arithmetic into fixtures -- run on any box that has the packages:

    pip install scikit-image opencv-python-headless pyiqa      # (none of them is in the build image: no package index there)
    python tests/golden/make_thirdparty_golden.py              # writes tests/golden/thirdparty_*.{json,npz} for what imports
    python -m pytest tests/test_thirdparty_pins.py -q          # oracle (CPU) and, with -m gpu on an MI355X, the HIP kernels

Each section runs only when its package imports; a fixture holds the package's output for the seeded inputs of
tests/thirdparty_refs.py (input digests stored, not the inputs), the package version, and nothing of the package itself.
With the fixtures committed, SURVEY rows a22 / a27 / a28 / a29 / 8f-3 / 8f-4 are pinned on boxes WITHOUT the packages too.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import thirdparty_refs as tp      # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    wrote = []
    if tp.have('skimage.metrics'):
        import skimage
        rows = []
        for name, img, ref in tp.image_pairs():
            rows.append({'name': name, 'img_sha': sha(img), 'ref_sha': sha(ref), 'mse': tp.skimage_mse(img, ref), 'ssim': tp.skimage_ssim(img, ref)})
        json.dump({'package': 'scikit-image ' + skimage.__version__, 'rows': rows}, open(os.path.join(HERE, 'thirdparty_metrics.json'), 'w'), indent=1)
        wrote.append('thirdparty_metrics.json')
    modes = [m for m, need in (('global', ['skimage']), ('local', ['skimage']), ('clahe', ['skimage', 'cv2'])) if tp.have(*need)]
    if modes:
        arrs, meta = {}, []
        for name, img in tp.histeq_images():
            for mode in modes:
                if mode == 'local' and img.size > 72 * 88:      # (disk(55) on a full frame: minutes in the numpy oracle -- the small image covers it)
                    continue
                arrs[f'{name}.{mode}'] = tp.thirdparty_histeq(img, mode).astype(np.float32)
                meta.append({'name': name, 'mode': mode, 'img_sha': sha(img)})
        arrs['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, 'thirdparty_histeq.npz'), **arrs)
        wrote.append('thirdparty_histeq.npz')
    if tp.have('cv2'):
        import cv2
        arrs, meta = {}, []
        for name, planes, gray in tp.color_inputs():
            arrs[name] = tp.cv2_color_merge(planes, gray)
            meta.append({'name': name, 'planes_sha': sha(planes), 'gray_sha': sha(gray), 'package': 'opencv ' + cv2.__version__})
        arrs['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, 'thirdparty_color.npz'), **arrs)
        wrote.append('thirdparty_color.npz')
    if tp.have('pyiqa'):
        try:
            import pyiqa
            run, sd = tp.pyiqa_lpips()
            img, ref = tp.lpips_pairs()
            json.dump({'package': 'pyiqa ' + getattr(pyiqa, '__version__', '?'), 'img_sha': sha(img), 'ref_sha': sha(ref),
                       'weights_sha': sha(np.concatenate([np.asarray(sd[k]).ravel() for k in sorted(sd)])),
                       'scores': [float(v) for v in run(img, ref)]}, open(os.path.join(HERE, 'thirdparty_lpips.json'), 'w'), indent=1)
            wrote.append('thirdparty_lpips.json')
        except Exception as e:      # no weights on disk and no network
            print('pyiqa imports but its LPIPS weights are unavailable:', e)
    print('wrote:', ', '.join(wrote) if wrote else 'nothing (no third-party package imports here)')


if __name__ == '__main__':
    main()

"""Build the 512x432 default-pattern fixture used by the synthetic benchmark/test scenes.

Run ONCE in the build container (needs /root/reference/data/default_pattern.png and PIL):

    python depthinspace_amd/data/make_pattern_fixture.py

What it does ("Pattern" row):
  * decode the 4096x4096 dot pattern, scale to [0,1]
  * apply the reference's orientation change for the default pattern, rot90(flip(axis=1))
    (reference data/data_manipulation.py:61-69)
  * bilinear remap into the 512x432 camera through K_proj * K^-1
    (reference data/create_syn_data.py:299-328): x' = s*(x-216)+2047.5, y' = s*(y-256)+2047.5,
    s = 1582.06005876/435.2
The output is a data fixture (float16, one channel, the three reference channels are identical).
"""
import os
import numpy as np


def build(src_png):
    from PIL import Image
    pat = np.asarray(Image.open(src_png)).astype(np.float32) / 255.0
    pat = np.rot90(np.flip(pat, axis=1))
    H, W = 512, 432
    s = 1582.06005876 / 435.2
    xs = (s * (np.arange(W, dtype=np.float64) - 216.0) + 2047.5)
    ys = (s * (np.arange(H, dtype=np.float64) - 256.0) + 2047.5)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    fx = (xs - x0).astype(np.float32)[None, :]
    fy = (ys - y0).astype(np.float32)[:, None]
    x1 = np.clip(x0 + 1, 0, pat.shape[1] - 1)
    y1 = np.clip(y0 + 1, 0, pat.shape[0] - 1)
    x0 = np.clip(x0, 0, pat.shape[1] - 1)
    y0 = np.clip(y0, 0, pat.shape[0] - 1)
    p00 = pat[y0][:, x0]
    p01 = pat[y0][:, x1]
    p10 = pat[y1][:, x0]
    p11 = pat[y1][:, x1]
    out = (p00 * (1 - fx) + p01 * fx) * (1 - fy) + (p10 * (1 - fx) + p11 * fx) * fy
    return out.astype(np.float16)


def build_real(src_png):
    """The `real` pattern as the reference's settings.pkl holds it (BASELINE config 5): no rotation / flip
    (reference data/data_manipulation.py:61-69 applies them to the default pattern only), projector and camera share
    K so the remap of data/create_syn_data.py:315-328 is the identity, then `post_process('real', ...)`
    (data/data_manipulation.py:91-105): crop [128:-128, 108:-108] -> 1024x864 and cv2.resize(INTER_LINEAR) to 512x432,
    which at an exact factor of 2 is the mean of each 2x2 block.  Channel mean is taken here (the workers only ever use
    pattern.mean(axis=2), reference model/multi_frame_worker.py:57-59)."""
    from PIL import Image
    pat = np.asarray(Image.open(src_png)).astype(np.float32) / 255.0
    if pat.ndim == 3:
        pat = pat.mean(axis=2)
    assert pat.shape == (1280, 1080), pat.shape
    pat = pat[128:-128, 108:-108]
    out = 0.25 * (pat[0::2, 0::2] + pat[0::2, 1::2] + pat[1::2, 0::2] + pat[1::2, 1::2])
    assert out.shape == (512, 432)
    return out.astype(np.float16)


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    out = build('/root/reference/data/default_pattern.png')
    np.savez_compressed(os.path.join(here, 'default_pattern_512x432.npz'), pattern=out)
    print(out.shape, out.dtype, float(out.min()), float(out.max()), float((out > 0.5).mean()))
    out = build_real('/root/reference/data/real_pattern.png')
    np.savez_compressed(os.path.join(here, 'real_pattern_512x432.npz'), pattern=out)
    print('real', out.shape, out.dtype, float(out.min()), float(out.max()), float((out > 0.5).mean()))

"""An independent anchor for the restated tf.image.resize(bicubic) (oracle/ops_ref.py; TensorFlow itself is not importable here, so the
kernel's semantics - Keys cubic a = -0.5, half-pixel centres, 1024-entry coefficient table, out-of-image taps dropped and the rest
renormalised - are restated from resize_bicubic_op.cc): Pillow's float-mode BICUBIC is another implementation of the same filter for
UPSCALING (same a, same pixel-centre convention, window clipped at the border and renormalised), without TF's table.  On white noise -
the worst case for an interpolator - the two must agree to the table's quantisation (offsets rounded to 1/1024: < 0.25 of a 0-255
range), exactly where every offset is a table entry (200 -> 256: offsets are multiples of 1/32), and bit-for-bit for the identity."""
import numpy as np
import pytest
import torch
from PIL import Image

from oracle import ops_ref as R


@pytest.mark.parametrize("h,w,oh,ow,tol", [(200, 200, 224, 224, 0.25), (200, 200, 256, 256, 1e-4), (200, 200, 384, 384, 0.25),
                                           (200, 200, 396, 396, 0.25), (37, 53, 64, 80, 0.25), (200, 200, 200, 200, 0.0)])
def test_oracle_bicubic_vs_pillow_float_bicubic(h, w, oh, ow, tol):
    rng = np.random.default_rng(h * 1000 + oh)
    img = (rng.random((h, w)) * 255).astype(np.float32)
    ours = R.resize_bicubic(torch.from_numpy(img)[..., None], oh, ow)[..., 0].numpy()
    pil = np.asarray(Image.fromarray(img, mode="F").resize((ow, oh), Image.BICUBIC))
    d = np.abs(ours - pil)
    assert d.max() <= tol, (d.max(), np.unravel_index(d.argmax(), d.shape))
    assert np.abs(ours[0] - pil[0]).max() <= tol and np.abs(ours[:, -1] - pil[:, -1]).max() <= tol     # the renormalised border taps

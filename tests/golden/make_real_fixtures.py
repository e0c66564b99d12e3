"""Cut the real-image fixtures (tests/golden/real/*.png) out of the reference's data files — run once, in the authoring
container where /root/reference is mounted:

    python tests/golden/make_real_fixtures.py && python tests/golden/make_golden.py

The reference keeps no test fixtures, but its tree holds leftovers of real runs: 21 input photographs
`temp_streamlit*/**/temp_sahi_input.jpg` (JPEG-decoded here with PIL, as sahi's read_image_as_pil does) and the face crops its
real detector + save_face_crops wrote (`.../crops/*_face_{i}_conf_{c}.jpg`: exactly what the enhancer is fed in
pipeline_v1). These are DATA (pixels), stored losslessly as small PNGs; no reference source travels. They put real JPEG
statistics — sensor noise, block edges, skin texture — through the fixed-point LetterBox resize, the stem and the SR net,
which the smooth synthetic frames never do. They do not pin the oracle to the reference's outputs (no weights: parity unpinned).
"""
import os

import numpy as np
from PIL import Image, ImageOps

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "real")

DET = {   # name: (file, x0, y0, w, h) — regions with faces / fine texture
    "det_test1": ("temp_streamlit/image_test_1_jpg/temp_sahi_input.jpg", 400, 180, 256, 256),
    "det_parade": ("temp_streamlit/image_0_Parade_marchingband_1_465_jpg/temp_sahi_input.jpg", 380, 150, 256, 256),
    "det_family": ("temp_streamlit/image_20_Family_Group_Family_Group_20_15_jpg/temp_sahi_input.jpg", 300, 120, 320, 224),
    "det_photo4k": ("temp_streamlit/image_foto abel_jpg/temp_sahi_input.jpg", 1900, 900, 256, 192),
}
SR = {    # whole face crops as the reference's own run saved them
    "sr_face_20x28": "temp_streamlit/image_20_Family_Group_Family_Group_20_15_jpg/crops/20_Family_Group_Family_Group_20_15.jpg_face_10_conf_0.82.jpg",
    "sr_face_23x27": "temp_streamlit/image_20_Family_Group_Family_Group_20_15_jpg/crops/20_Family_Group_Family_Group_20_15.jpg_face_11_conf_0.82.jpg",
    "sr_face_47x54": "temp_streamlit/image_20_Family_Group_Family_Group_20_15_jpg/crops/20_Family_Group_Family_Group_20_15.jpg_face_10_conf_0.83.jpg",
}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, (f, x0, y0, w, h) in DET.items():
        im = ImageOps.exif_transpose(Image.open(os.path.join(REF, f))).convert("RGB")
        im.crop((x0, y0, x0 + w, y0 + h)).save(os.path.join(OUT, name + ".png"), optimize=True)
    for name, f in SR.items():
        Image.open(os.path.join(REF, f)).convert("RGB").save(os.path.join(OUT, name + ".png"), optimize=True)
    for f in sorted(os.listdir(OUT)):
        a = np.asarray(Image.open(os.path.join(OUT, f)))
        print(f, a.shape, os.path.getsize(os.path.join(OUT, f)), "bytes, pixel std %.1f" % a.std())


if __name__ == "__main__":
    main()

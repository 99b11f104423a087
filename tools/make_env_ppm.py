"""Decodes a scene's environment.jpg to the binary PPM the C++ loader reads.

The reference decodes the JPEG at load time with the `image` crate
(src/scene.rs:120-124).  No JPEG decoder is linked into the C++ host, so the
decoded RGB8 texels are committed next to the scene files instead:

    python tools/make_env_ppm.py /root/reference/scene/buddha/environment.jpg scenes/buddha/environment.ppm

Decoders may differ by +-1 LSB per texel; that is far below the render noise.
"""
import sys

from PIL import Image


def main(src, dst):
    im = Image.open(src).convert("RGB")
    with open(dst, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % im.size)
        f.write(im.tobytes())
    print(dst, im.size)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

"""jpeg-encoder_amd — MI355X-native JPEG block-encode pipeline (host-side Python surface).

The product is the C-ABI shared library built from csrc/ (see include/jpegenc_mi355x.h); this
package only holds the ctypes binding used by tests and bench.py, and synthetic-input helpers.
The directory name contains a hyphen, so import it through `__graft_entry__.load_package()`,
which registers it as the module `jpeg_encoder_amd`.
"""
__all__ = ["synth"]

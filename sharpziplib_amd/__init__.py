"""sharpziplib_amd — MI355X-native DEFLATE engine behind SharpZipLib's Deflater/Inflater API.

Product code lives in `csrc/` (HIP kernels + C ABI, built into libszl_amd.so) and in the
host-side mirrors of the reference's four boundary classes (`deflater.py`, `inflater.py`,
`streams.py`).  Nothing here imports `oracle/`.
"""
__version__ = "0.1.0"

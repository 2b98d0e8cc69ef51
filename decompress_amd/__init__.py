"""decompress_amd -- MI355X-native batched decode engine behind the
se-mz/decompress interface (decompress / decompress-all / format keywords).

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI) and the
host-side mirror of the reference's public interface (api.py).
"""
__version__ = "0.1.0"

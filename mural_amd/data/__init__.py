from .genome import PackedGenome, pack_sequence  # noqa: F401

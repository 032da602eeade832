from .genome import PackedGenome, SymbolWindows, pack_sequence  # noqa: F401
from .ingest import read_fasta, read_bed, bed_order, predict_bed, scan_fasta, pack_fasta_record, poisson_calibrate, write_predictions, packed_segments, train_batches_from_files  # noqa: F401

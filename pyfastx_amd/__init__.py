"""pyfastx_amd -- MI355X-native FASTA/FASTQ index build and random access
behind the pyfastx object API (Fasta / Fastq / Sequence / Read)."""
__version__ = "0.1.0"

from .api import Fasta, Fastq, Fastx, Sequence, Read, FastaKeys, FastqKeys, version, gzip_check, reverse_complement  # noqa: E402,F401

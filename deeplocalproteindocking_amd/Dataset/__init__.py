"""``from Dataset import get_benchmark_stream`` as the reference's local_test.py:8 does.  The
training stream (SplitComplexDataset) is outside the docking path (out of scope)."""
from .SplitComplexBenchmark import get_benchmark_stream, read_pdb_list, read_dataset_list, SplitComplexBenchmark


def get_dataset_stream(*args, **kwargs):
    raise Exception("Dataset.get_dataset_stream: training data loading is out of scope of this build")

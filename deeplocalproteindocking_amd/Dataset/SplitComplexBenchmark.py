"""DockingBenchmark target table -> the 7-tuples ``local_test.py`` iterates over
(SURVEY.md 8(f) row 3).  Behaviour of /root/reference/src/Dataset/SplitComplexBenchmark.py:

  read_pdb_list (:20-46)      Table_BM5-style tab-separated file; rows before the first section
                              header are ignored; section headers set the difficulty class
                              (1 rigid-body, 2 medium, 3 difficult) and are not targets themselves
  read_dataset_list (:48-64)  debug listing: names + Description/<name>.dat (2nd line: receptor ligand)
  SplitComplexBenchmark (:67-109), get_benchmark_stream (:112-115): batch-1, unshuffled DataLoader

Pinned by tests/golden/g6_consumers.npz (captured from the reference's own functions).
"""
import os

import torch
from torch.utils.data import Dataset

_SECTIONS = (("Rigid-body", 1), ("Medium Difficulty", 2), ("Difficult", 3))


def _section_of(line):
    # checked in this order: "Medium Difficulty" contains "Difficult"
    for key, cls in _SECTIONS:
        if key in line:
            return cls
    return 0


def read_pdb_list(benchmark_dir, pdb_list_file, struct_folder="structures"):
    targets, cls = [], 0
    with open(pdb_list_file) as fin:
        for line in fin:
            sec = _section_of(line)
            if sec:
                cls = sec
                continue
            if cls == 0:
                continue
            # (the reference appends a target for EVERY line of a section, an empty one included -- its name is then the
            #  line break; such rows, and the CR of a CRLF table, are dropped here instead of becoming file names)
            pdb = line.split("\t")[0].split("_")[0].strip()
            if not pdb:
                continue
            s = lambda suffix: os.path.join(benchmark_dir, struct_folder, pdb + suffix)
            targets.append((pdb, os.path.join(benchmark_dir, "Matched", pdb + "_b.pdb"),
                            s("_r_u.pdb"), s("_l_u.pdb"), s("_r_b.pdb"), s("_l_b.pdb"), cls))
    return targets


def read_dataset_list(benchmark_dir, pdb_list_file):
    targets = []
    with open(pdb_list_file) as fin:
        for line in fin:
            pdb = line.split()[0]
            with open(os.path.join(benchmark_dir, "Description", pdb + ".dat")) as desc:
                desc.readline()
                receptor, ligand = desc.readline().split()[:2]
            targets.append((pdb, os.path.join(benchmark_dir, "Structures", pdb + ".pdb"),
                            receptor, ligand, receptor, ligand, 0))
    return targets


class SplitComplexBenchmark(Dataset):
    def __init__(self, dataset_dir, struct_folder="structures", description_set="Table_BM5.csv", debug=False):
        self.dataset_dir = dataset_dir
        if debug:
            self.targets = read_dataset_list(dataset_dir, os.path.join(dataset_dir, "Description", description_set))
        else:
            self.targets = read_pdb_list(dataset_dir, os.path.join(dataset_dir, description_set),
                                         struct_folder=struct_folder)
        self.dataset_size = len(self.targets)
        print("Dataset file: ", self.dataset_dir)
        print("Dataset size: ", self.dataset_size)

    def __getitem__(self, index):
        return self.targets[index]

    def __len__(self):
        return self.dataset_size


def get_benchmark_stream(data_dir, struct_folder="structures", subset="Table_BM5.csv", debug=False):
    dataset = SplitComplexBenchmark(data_dir, struct_folder=struct_folder, description_set=subset, debug=debug)
    return torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False, num_workers=0)

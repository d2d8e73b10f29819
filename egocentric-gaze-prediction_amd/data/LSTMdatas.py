"""Mirror of the reference's ``data/LSTMdatas.py``: consecutive files of extracted 512-vectors; sample i =
(file[i], file[i+1], same-video flag) (data/LSTMdatas.py:44-59)."""
import os

import torch
from torch.utils.data import Dataset


class lstmDataset(Dataset):
    def __init__(self, Path='../512w/test', name=None):
        self.Path = Path
        files = os.listdir(Path)
        self.listFiles = sorted(files if name is None else [k for k in files if name in k])

    def __len__(self):
        return len(self.listFiles) - 1

    def __getitem__(self, index):
        a, b = self.listFiles[index], self.listFiles[index + 1]
        return {'input': torch.load(os.path.join(self.Path, a)), 'gt': torch.load(os.path.join(self.Path, b)),
                'same': b[:-14] == a[:-14]}

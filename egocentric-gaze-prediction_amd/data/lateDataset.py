"""Mirror of the reference's ``data/lateDataset.py``: (SP prediction, ground truth, AT feature map) triples as
u8/255 maps of shape (1,H,W) (data/lateDataset.py:9-34)."""
import os

import torch
from torch.utils.data import Dataset

from ._io import imread


class lateDataset(Dataset):
    def __init__(self, imgPath_s, gtPath, featPath, listFiles, listGtFiles, listFeat):
        self.imgPath_s, self.gtPath, self.featPath = imgPath_s, gtPath, featPath
        self.listFiles, self.listGtFiles, self.listFeat = listFiles, listGtFiles, listFeat

    def __len__(self):
        return len(self.listGtFiles)

    def _load(self, folder, name):
        return torch.from_numpy(imread(os.path.join(folder, name), gray=True)).float().div(255).unsqueeze(0)

    def __getitem__(self, index):
        return {'im': self._load(self.imgPath_s, self.listFiles[index]),
                'gt': self._load(self.gtPath, self.listGtFiles[index]),
                'feat': self._load(self.featPath, self.listFeat[index])}

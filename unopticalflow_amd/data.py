"""Inputs for the flow stage: frame triplets [B,3,3H,W] (left, centre, right stacked on H), BGR/255.

``SyntheticTriplets`` generates batches on the device (bench / smoke / --synthetic training).
``PreparedTriplets`` reads the stacked-triplet PNGs + train.txt that the reference's
``KITTI_RAW.prepare_data_mp`` / ``SINTEL_RAW.prepare_data_mp`` write (kitti_prepared.py:10-42,133-153)
with PIL instead of cv2 (resize to img_hw, random horizontal flip, /255, channel order BGR so the
published checkpoints see what they were trained on).  Image decoding stays on the CPU: it is an
I/O stage outside the kernel scope.
"""
import os

import numpy as np
import torch
import torch.utils.data


class SyntheticTriplets:
    def __init__(self, batch_size, img_hw, device, seed=0):
        self.shape = (batch_size, 3, 3 * img_hw[0], img_hw[1])
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        return torch.rand(self.shape, generator=self.gen, device=self.device, dtype=torch.float32)


class PreparedTriplets(torch.utils.data.Dataset):
    def __init__(self, data_dir, num_scales=3, img_hw=(256, 832), num_iterations=None):
        self.data_dir, self.num_scales, self.img_hw, self.num_iterations = data_dir, num_scales, img_hw, num_iterations
        with open(os.path.join(data_dir, 'train.txt')) as f:
            self.files = [os.path.join(data_dir, ln.split()[0]) for ln in f if ln.strip()]
        print('A total of {} image pairs found'.format(len(self.files)))

    def count(self):
        return len(self.files)

    def __len__(self):
        return self.count() if self.num_iterations is None else self.num_iterations

    def rand_num(self, idx):                       # kitti_prepared.py:38-42
        np.random.seed(idx)
        return np.random.randint(self.count())

    def __getitem__(self, idx):
        from PIL import Image
        if self.num_iterations is not None:
            idx = self.rand_num(idx)
        img = Image.open(self.files[idx]).convert('RGB')
        w, h3 = img.size
        h = h3 // 3
        frames = [img.crop((0, k * h, w, (k + 1) * h)).resize((self.img_hw[1], self.img_hw[0]), Image.BILINEAR)
                  for k in range(3)]
        arr = np.concatenate([np.asarray(f) for f in frames], 0)[:, :, ::-1]     # RGB -> BGR (cv2 order)
        if np.random.rand() > 0.5:
            arr = arr[:, ::-1]
        arr = np.ascontiguousarray(arr.transpose(2, 0, 1)).astype(np.float32) / 255.0
        return torch.from_numpy(arr)

"""Label files exactly as /root/reference/scripts/prepro_labels.py:158-163 writes them (h5py.File(path, 'w') + create_dataset(name,
dtype='uint32', data=...)), for the h5py-free reader captioning/data/h5lite.py.  Needs an interpreter with h5py (the build
container has one at /opt/conda/bin/python3.9; the product and its tests never import it):

    /opt/conda/bin/python3.9 tests/golden/make_h5.py

labels_small.h5: 7 images; labels_many.h5: 300 datasets-worth of names is not needed, but 3000 images make the datasets span
many KB and `labels` 2-D; labels_chunked.h5: one chunked + compressed dataset the reader must refuse with a clear message."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def write(name, n_img, seed, width=16):
    rng = np.random.default_rng(seed)
    per = rng.integers(1, 7, size=n_img)
    M = int(per.sum())
    length = rng.integers(1, width + 1, size=M).astype('uint32')
    L = np.zeros((M, width), dtype='uint32')
    for i, ln in enumerate(length):
        L[i, :ln] = rng.integers(1, 9488, size=ln)
    end = np.cumsum(per).astype('uint32')
    start = (end - per + 1).astype('uint32')
    f = h5py.File(os.path.join(HERE, name), 'w')
    f.create_dataset('labels', dtype='uint32', data=L)
    f.create_dataset('label_start_ix', dtype='uint32', data=start)
    f.create_dataset('label_end_ix', dtype='uint32', data=end)
    f.create_dataset('label_length', dtype='uint32', data=length)
    f.close()
    np.savez_compressed(os.path.join(HERE, name.replace('.h5', '_expected.npz')), labels=L, label_start_ix=start, label_end_ix=end,
                        label_length=length)


write('labels_small.h5', 7, 0)
write("labels_many.h5", 500, 1)
f = h5py.File(os.path.join(HERE, 'labels_chunked.h5'), 'w')
f.create_dataset('labels', data=np.arange(4096, dtype='uint32').reshape(256, 16), chunks=(64, 16), compression='gzip')
f.close()
print('written', [n for n in os.listdir(HERE) if n.startswith('labels_')])

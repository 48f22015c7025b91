#!/bin/bash
# End-to-end tools/train.py on files of the BASELINE shape (UpDown SCST bs10 x 5 over a 300-image on-disk dataset, 36 x 2048 regions,
# savez_compressed): loader + prefetcher + step + per-iteration loss.item(), with the HBM-resident store and with streaming.
#   scripts/train_e2e.sh [iterations]
it=${1:-240}
d=$(mktemp -d)
python - "$d" <<'PY'
import json, os, sys
import numpy as np
d = sys.argv[1]
os.mkdir(d + '/att')
rng = np.random.default_rng(0)
labels, start, end, images = [], [], [], []
for i in range(300):
    np.savez_compressed('%s/att/%d.npz' % (d, i), feat=np.clip(rng.standard_normal((36, 2048)), 0, None).astype(np.float32))
    start.append(len(labels) + 1)
    for _ in range(5):
        row = np.zeros(20, dtype=np.uint32)
        n = int(rng.integers(6, 18))
        row[:n] = rng.integers(1, 9488, size=n)
        labels.append(row)
    end.append(len(labels))
    images.append({'id': i, 'split': 'train'})
json.dump({'images': images, 'ix_to_word': {str(i): 'w%d' % i for i in range(1, 9488)}}, open(d + '/d.json', 'w'))
np.savez(d + '/l.npz', labels=np.stack(labels), label_start_ix=np.array(start, dtype=np.uint32), label_end_ix=np.array(end, dtype=np.uint32))
PY
for res in ${MODES:-1 0 0p}; do
  echo "resident_features=$res   (0p: streaming with decode worker processes)"
  procs=0; [ $res = 0p ] && procs=1
  CAPMI_LOADER_PROCS=$procs python -m imagecaptioning.pytorch_amd.tools.train --caption_model updown --rnn_size 1000 --input_encoding_size 1000 --att_hid_size 512 --input_json $d/d.json --input_label_h5 $d/l.npz \
     --input_att_dir $d/att --batch_size 10 --seq_per_img 5 --self_critical_after 0 --train_sample_n 5 --max_iters $it --max_epochs -1 \
     --losses_log_every 1000 --resident_features ${res%p} 2>&1 | grep -v amdgpu.ids | tail -2
done
rm -rf $d

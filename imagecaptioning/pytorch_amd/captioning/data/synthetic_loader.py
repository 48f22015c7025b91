"""Synthetic stand-in for captioning/data/dataloader.py (h5py / lmdbdict loaders are outside the hot path and not
installable here).  It produces batches with EXACTLY the reference's batch-dict contract (dataloader.py:229-258):
``fc_feats [B,F]``, ``att_feats [B,K,F]``, ``att_masks`` (None when all images have K regions, :240-241),
``labels [B,n,L+2]`` int64 with BOS/EOS columns 0, ``masks [B,n,L+2]`` (nonzeros+2 ones, :245-249),
``gts`` list of uint32 [n_ref,L], ``bounds``, ``infos``."""
import numpy as np
import torch

from imagecaptioning.pytorch_amd import synthetic


class SyntheticLoader:
    def __init__(self, opt, rank=0, world=1):
        """rank / world: the synthetic images are dealt out to the data-parallel ranks (image i belongs to rank i % world), so
        the ranks of a pass see disjoint images of ONE corpus (same `seed`, hence the same document-frequency table everywhere)"""
        self.opt = opt
        self.rank, self.world = int(rank), max(1, int(world))
        self.batch_size = opt.batch_size
        self.seq_per_img = opt.seq_per_img
        self.seq_length = opt.seq_length
        self.vocab_size = opt.vocab_size
        self.ix_to_word = {str(i): ('w%d' % i) for i in range(1, self.vocab_size)}
        self.ix_to_word[str(self.vocab_size)] = 'UNK'
        rng = np.random.default_rng(opt.seed)
        n_img = opt.synthetic_images
        self.refs = [synthetic.zipf_rows(rng, 5, self.seq_length, vocab=self.vocab_size) for _ in range(n_img)]
        self.seeds = rng.integers(0, 2 ** 31, size=n_img)
        # this rank's share of a pass: every world-th image, the tail padded from the head of the pass (as FeatureLoader._mine
        # does) so that all ranks hold ceil(n / world) images and wrap -- advance the epoch, hence lr decay / ss_prob / the
        # XE -> SCST switch -- on the same batch
        share = -(-n_img // self.world)
        self.mine = [(self.rank + self.world * j) % n_img for j in range(share)]
        self.pos = {'train': 0, 'val': 0, 'test': 0}
        self.epoch_wrapped = False

    def get_vocab(self):
        return self.ix_to_word

    def document_frequency(self):
        return synthetic.document_frequency(self.refs)

    def reset_iterator(self, split):
        """DataLoader.reset_iterator (dataloader.py:356-358)"""
        self.pos[split] = 0

    def get_batch(self, split, batch_size=None):
        B = batch_size or self.batch_size
        n, L, opt = self.seq_per_img, self.seq_length, self.opt
        mine = len(self.mine)                                          # images of this rank (equal on every rank)
        idx = [self.mine[(self.pos[split] + i) % mine] for i in range(B)]
        wrapped = self.pos[split] + B >= mine
        self.pos[split] = (self.pos[split] + B) % mine
        fc = np.zeros((B, opt.fc_feat_size), dtype=np.float32)
        att = np.zeros((B, opt.synthetic_regions, opt.att_feat_size), dtype=np.float32)
        labels = np.zeros((B, n, L + 2), dtype=np.int64)
        masks = np.zeros((B, n, L + 2), dtype=np.float32)
        gts, infos = [], []
        for b, ix in enumerate(idx):
            g = np.random.default_rng(int(self.seeds[ix]))
            att[b] = np.clip(g.standard_normal(att[b].shape) * 0.5, 0, None)
            fc[b] = att[b].mean(0)
            rows = self.refs[ix][g.integers(0, 5, size=n)]
            labels[b, :, 1:L + 1] = rows
            for j in range(n):
                masks[b, j, :int((rows[j] > 0).sum()) + 2] = 1
            gts.append(self.refs[ix])
            infos.append({'ix': ix, 'id': ix, 'file_path': 'synthetic/%d' % ix})
        return {'fc_feats': torch.from_numpy(fc), 'att_feats': torch.from_numpy(att), 'att_masks': None,
                'labels': torch.from_numpy(labels), 'masks': torch.from_numpy(masks), 'gts': gts,
                'bounds': {'it_pos_now': self.pos[split], 'it_max': mine, 'wrapped': wrapped}, 'infos': infos}

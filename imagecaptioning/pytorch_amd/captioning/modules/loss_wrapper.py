"""LossWrapper (reference captioning/modules/loss_wrapper.py:18-75): picks XE / SCST / structure
loss per step.  Same constructor, call signature and output dict.  Differences are all on the
device side: the SCST reward never leaves HBM (no .cpu().numpy() round trip, rewards.py:48-49)."""
import torch

from . import losses
from ..utils.rewards import self_critical_reward_device, select_gts


class LossWrapper(torch.nn.Module):
    def __init__(self, model, opt):
        super().__init__()
        self.opt = opt
        self.model = model
        if getattr(opt, 'label_smoothing', 0) > 0:
            self.crit = losses.LabelSmoothing(smoothing=opt.label_smoothing)
        else:
            self.crit = losses.LanguageModelCriterion()
        self.rl_crit = losses.RewardCriterion()
        self.struc_crit = losses.StructureLosses(opt) if getattr(opt, 'structure_loss_type', None) else None

    def forward(self, fc_feats, att_feats, labels, masks, att_masks, gts, gt_indices, sc_flag, struc_flag,
                drop_worst_flag):
        opt = self.opt
        out = {}
        reduction = 'none' if drop_worst_flag else 'mean'
        if struc_flag:
            if getattr(opt, 'use_ppo', 0):
                raise NotImplementedError('PPO loss is out of scope (SURVEY.md 2.1 #12)')
            w = opt.structure_loss_weight
            if w < 1:
                lm_loss = self.crit(self.model(fc_feats, att_feats, labels[..., :-1], att_masks), labels[..., 1:],
                                    masks[..., 1:], reduction=reduction)
            else:
                # (a device fill: the reference's torch.tensor(0).type_as(fc_feats) is a blocking host -> device copy)
                lm_loss = (att_feats if att_feats is not None else fc_feats).new_zeros(())
            if w > 0:
                gen_result, sample_logprobs = self.model(
                    fc_feats, att_feats, att_masks,
                    opt={'sample_method': opt.train_sample_method, 'beam_size': opt.train_beam_size,
                         # loss_wrapper.py:34-35: the margin losses (softmax_margin excepted) read raw logits
                         'output_logsoftmax': int(bool(getattr(opt, 'struc_use_logsoftmax', False))
                                                  or opt.structure_loss_type == 'softmax_margin'
                                                  or 'margin' not in opt.structure_loss_type),
                         'sample_n': opt.train_sample_n}, mode='sample')
                gts = select_gts(gts, gt_indices)
                struc_loss = self.struc_crit(sample_logprobs, gen_result, gts, reduction=reduction)
            else:
                z = (att_feats if att_feats is not None else fc_feats).new_zeros(())
                struc_loss = {'loss': z, 'reward': z}
            loss = (1 - w) * lm_loss + w * struc_loss['loss']
            out['lm_loss'] = lm_loss
            out['struc_loss'] = struc_loss['loss']
            out['reward'] = struc_loss['reward']
        elif not sc_flag:
            loss = self.crit(self.model(fc_feats, att_feats, labels[..., :-1], att_masks), labels[..., 1:], masks[..., 1:],
                             reduction=reduction)
        else:
            fused = (getattr(self, 'fuse_scst_rollouts', True) and hasattr(self.model, 'scst_rollouts')
                     and opt.sc_sample_method == 'greedy' and opt.sc_beam_size == 1
                     and opt.train_sample_method == 'sample' and opt.train_beam_size == 1)
            if fused:
                # greedy baseline + sampled rollouts in one pass over the weights (see AttModel.scst_rollouts)
                self.model.train()
                greedy_res, gen_result, sample_logprobs = self.model.scst_rollouts(
                    fc_feats, att_feats, att_masks, sample_n=opt.train_sample_n)
            else:
                self.model.eval()
                with torch.no_grad():
                    greedy_res, _ = self.model(fc_feats, att_feats, att_masks, mode='sample',
                                               opt={'sample_method': opt.sc_sample_method, 'beam_size': opt.sc_beam_size})
                self.model.train()
                gen_result, sample_logprobs = self.model(
                    fc_feats, att_feats, att_masks,
                    opt={'sample_method': opt.train_sample_method, 'beam_size': opt.train_beam_size,
                         'sample_n': opt.train_sample_n}, mode='sample')
            gts = select_gts(gts, gt_indices)
            adv, _scores = self_critical_reward_device(greedy_res, gts, gen_result, opt)      # [N] on device
            reward = adv.unsqueeze(1).expand(-1, gen_result.shape[1])
            loss = self.rl_crit(sample_logprobs, gen_result.data, reward, reduction=reduction)
            out['reward'] = getattr(adv, '_capmi_mean', None) if getattr(adv, '_capmi_mean', None) is not None else adv.mean()
        out['loss'] = loss
        return out

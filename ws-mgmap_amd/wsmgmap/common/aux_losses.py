"""Process-global registry of per-sample auxiliary losses (the trainer activates / clears
it around every update).  Mirrors the interface of the reference's
vlnce_baselines/common/aux_losses.py:4-47 (`AuxLosses` singleton: clear, register_loss,
get_loss, reduce, is_active, activate, deactivate) and its error behaviour (asserts)."""
import torch


class _Registry:
    def __init__(self):
        self._entries = {}  # name -> (per-sample loss [B], alpha)
        self._on = False
        self._alpha_key, self._alpha = None, None   # device copy of the alphas (they are configuration constants)

    def activate(self):
        self._on = True

    def deactivate(self):
        self._on = False

    def is_active(self):
        return self._on

    def clear(self):
        self._entries.clear()

    def register_loss(self, name, loss, alpha=1.0):
        assert self._on
        assert name not in self._entries
        self._entries[name] = (loss, alpha)

    def get_loss(self, name):
        return self._entries[name][0]

    def reduce(self, mask=None):
        """sum_k alpha_k * mean(loss_k[mask])"""
        assert self._on
        entries = list(self._entries.values())
        if mask is None or not entries or not all(l.dim() == 1 and l.shape == entries[0][0].shape for l, _ in entries):
            total = 0.0
            for loss, alpha in entries:
                if mask is None:
                    total = total + alpha * loss.mean()
                else:
                    # = masked_select(loss, mask).mean(): rows outside the mask are DROPPED (torch.where), not multiplied
                    # by 0 — a non-finite loss on a padded row (log of an underflowed attention weight) must not
                    # poison the total
                    total = total + alpha * (torch.where(mask, loss, loss.new_zeros(())).sum() / mask.sum())
            return total
        first = entries[0][0]
        if first.is_cuda and len(entries) <= 4 and mask.dtype == torch.bool and all(l.dtype == torch.float32 for l, _ in entries):
            # the same sum as below in one launch per direction (csrc/wsmg_heads.hip): selected rows only, NaN for an empty selection
            from .. import ops
            return ops.aux_reduce([l for l, _ in entries], [a for _, a in entries], mask)
        # = sum_k alpha_k * masked_select(loss_k, mask).mean() (NaN for an empty selection, like the reference), without
        # the data-dependent output size — masked_select makes the host wait for the whole forward pass — and as ONE
        # stacked reduction: per loss the loop above is 7 tiny launches forward and as many backward, each ~5 us on the
        # critical path between the two recurrences' forward and backward
        first = entries[0][0]
        key = (tuple(float(a) for _, a in entries), first.device, first.dtype)
        if self._alpha_key != key:
            self._alpha_key, self._alpha = key, torch.tensor(key[0], device=first.device, dtype=first.dtype)
        stacked = torch.stack([l for l, _ in entries])            # [K, B]
        kept = torch.where(mask.unsqueeze(0), stacked, stacked.new_zeros(()))   # drop (not zero-multiply) masked-out rows
        return ((kept.sum(dim=1) * self._alpha).sum()) / mask.sum()


AuxLosses = _Registry()

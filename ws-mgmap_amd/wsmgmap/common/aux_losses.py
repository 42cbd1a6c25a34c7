"""Process-global registry of per-sample auxiliary losses (the trainer activates / clears
it around every update).  Mirrors the interface of the reference's
vlnce_baselines/common/aux_losses.py:4-47 (`AuxLosses` singleton: clear, register_loss,
get_loss, reduce, is_active, activate, deactivate) and its error behaviour (asserts)."""
import torch


class _Registry:
    def __init__(self):
        self._entries = {}  # name -> (per-sample loss [B], alpha)
        self._on = False

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
        total = 0.0
        for loss, alpha in self._entries.values():
            if mask is None:
                total = total + alpha * loss.mean()
            else:
                # = masked_select(loss, mask).mean() (NaN for an empty selection, like the reference), without the
                # data-dependent output size — masked_select makes the host wait for the whole forward pass
                m = mask.to(loss.dtype)
                total = total + alpha * ((loss * m).sum() / m.sum())
        return total


AuxLosses = _Registry()

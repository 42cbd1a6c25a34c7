"""Instruction encoder: embedding + packed bidirectional LSTM.  State_dict keys and config
fields follow the reference (instruction_encoder.py:10-93).  The recurrent cell itself is the
stock PyTorch-ROCm (MIOpen) LSTM — it is not one of the three hand-written operators.

Differences in data flow (results identical):
  * output is token-major [B, L, 256] (what the attention kernel streams), the reference's
    [B, 256, L] is a transpose view of it;
  * identical instructions in a batch (teacher-forcing batches repeat each episode's
    instruction over T time steps) are encoded once and gathered.
"""
import gzip
import json

import torch
import torch.nn as nn


class InstructionEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.use_pretrained_embeddings:
            with gzip.open(config.embedding_file, "rt") as f:
                table = torch.tensor(json.load(f))
            self.embedding_layer = nn.Embedding.from_pretrained(embeddings=table, freeze=not config.fine_tune_embeddings)
        else:
            self.embedding_layer = nn.Embedding(config.vocab_size, config.embedding_size, padding_idx=0)
        cell = nn.GRU if config.rnn_type == "GRU" else nn.LSTM
        self.bidir = config.bidirectional
        self.encoder_rnn = cell(input_size=config.embedding_size, hidden_size=config.hidden_size, bidirectional=self.bidir)
        self.final_state_only = config.final_state_only

    @property
    def output_size(self):
        return self.config.hidden_size * (2 if self.bidir else 1)

    @staticmethod
    def _dedup(tokens):
        """Unique rows of a [B, L] token matrix without a row-wise sort: a 64-bit polynomial hash per
        row, a 1-D unique over the B hashes, and an exact on-device check (a hash collision falls back
        to torch.unique(dim=0)).  Returns (unique rows [U, L], inverse [B], lengths on the host)."""
        B, L = tokens.shape
        mult = (torch.arange(1, L + 1, device=tokens.device, dtype=torch.int64) * 0x9E3779B97F4A7C15) | 1
        h = (tokens * mult).sum(dim=1)
        _, inverse = torch.unique(h, return_inverse=True)
        U = int(inverse.max()) + 1 if B else 0  # host sync (the packed LSTM needs host lengths anyway)
        rep = torch.full((U,), B, device=tokens.device, dtype=torch.int64)
        rep.scatter_reduce_(0, inverse, torch.arange(B, device=tokens.device), reduce="amin")
        uniq = tokens[rep]
        lengths = (uniq != 0).long().sum(dim=1)
        exact = (uniq[inverse] == tokens).all().view(1).long()
        host = torch.cat([exact, lengths]).cpu()
        if int(host[0]) != 1:  # hash collision: exact path
            uniq, inverse = torch.unique(tokens, dim=0, return_inverse=True)
            return uniq, inverse, (uniq != 0).long().sum(dim=1).cpu()
        return uniq, inverse, host[1:]

    def encode_unique(self, instruction):
        """-> (hidden [U, L, D] token-major, pad mask [U, L] bool, inverse [B]) with U unique rows."""
        tokens = instruction.long()
        uniq, inverse, lengths = self._dedup(tokens)
        embedded = self.embedding_layer(uniq)
        packed = nn.utils.rnn.pack_padded_sequence(embedded, lengths, batch_first=True, enforce_sorted=False)
        output, _ = self.encoder_rnn(packed)
        hidden = nn.utils.rnn.pad_packed_sequence(output, batch_first=True)[0]  # [U, L, D]
        mask = (hidden == 0.0).all(dim=2)
        return hidden.contiguous(), mask, inverse

    def forward(self, observations):
        """Reference-shaped result: ([B, D, L], mask [B, L])."""
        hidden, mask, inverse = self.encode_unique(observations["instruction"])
        return hidden[inverse].permute(0, 2, 1), mask[inverse]

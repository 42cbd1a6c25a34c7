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

    def encode_unique(self, instruction):
        """-> (hidden [U, L, D] token-major, pad mask [U, L] bool, inverse [B]) with U unique rows."""
        tokens = instruction.long()
        uniq, inverse = torch.unique(tokens, dim=0, return_inverse=True)
        lengths = (uniq != 0).long().sum(dim=1)
        embedded = self.embedding_layer(uniq)
        packed = nn.utils.rnn.pack_padded_sequence(embedded, lengths.cpu(), batch_first=True, enforce_sorted=False)
        output, _ = self.encoder_rnn(packed)
        hidden = nn.utils.rnn.pad_packed_sequence(output, batch_first=True)[0]  # [U, L, D]
        mask = (hidden == 0.0).all(dim=2)
        return hidden.contiguous(), mask, inverse

    def forward(self, observations):
        """Reference-shaped result: ([B, D, L], mask [B, L])."""
        hidden, mask, inverse = self.encode_unique(observations["instruction"])
        return hidden[inverse].permute(0, 2, 1), mask[inverse]

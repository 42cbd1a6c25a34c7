"""ORACLE (test infrastructure, never on the product path): functional PyTorch-CPU fp32
restatement of the reference's policy network.  `P` is a flat {state_dict key: tensor}
dict with the REFERENCE's key names; nothing here is an nn.Module tree.

Follows (all under /root/reference/vlnce_baselines/):
    models/policy.py:30-103                BasePolicy.update_map / act / aux_prediction / forward
    models/mg_map_policy.py:173-251        MGMapNet._attn / forward
    models/encoders/map_encoder.py:16-112  MapEncoder, MapDecoder (resnet18 stem: conv1,bn1,relu,maxpool,layer1)
    models/encoders/instruction_encoder.py:68-93
    models/encoders/unet_encoder.py:64-111 ResNetUNet (frozen, eval-mode BN)
    models/encoders/resnet_encoders.py:25-32,72-102 (DD-PPO GroupNorm ResNet50: 3p, restated; spatial embedding concat)
    common/distributions.py:42-57          DiagGaussian
    common/aux_losses.py:24-35             masked mean reduce
    dagger_trainer.py:526-533              action loss
Third-party pieces restated from their pinned versions (parity UNPINNED at those
boundaries): habitat-lab v0.1.5 RNNStateEncoder (GRU with h*mask),
CriticHead (Linear 512->1), torchvision BasicBlock.

Pinned against tests/golden/g3_update.npz, g4_act.npz, g5_attn.npz.
"""
import math

import torch
import torch.nn.functional as F

from . import bev_ref


# ----------------------------------------------------------------------------- small pieces
def _bn(P, pre, x, train):
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"],
                        training=train, momentum=0.1, eps=1e-5)


def _bump(P, pre, train):
    k = pre + ".num_batches_tracked"
    if train and k in P:
        P[k] += 1


def _conv(P, pre, x, stride=1, padding=0):
    return F.conv2d(x, P[pre + ".weight"], P.get(pre + ".bias"), stride=stride, padding=padding)


def _convrelu(P, pre, x, padding, train):
    """convrelu(): Conv2d(+bias) -> BatchNorm2d -> ReLU  (map_encoder.py:8-13)."""
    y = _conv(P, pre + ".0", x, 1, padding)
    _bump(P, pre + ".1", train)
    return F.relu(_bn(P, pre + ".1", y, train))


def _basic_block(P, pre, x, train, stride=1):
    idt = x
    y = _conv(P, pre + ".conv1", x, stride, 1)
    _bump(P, pre + ".bn1", train)
    y = F.relu(_bn(P, pre + ".bn1", y, train))
    y = _conv(P, pre + ".conv2", y, 1, 1)
    _bump(P, pre + ".bn2", train)
    y = _bn(P, pre + ".bn2", y, train)
    if (pre + ".downsample.0.weight") in P:
        idt = _conv(P, pre + ".downsample.0", x, stride, 0)
        _bump(P, pre + ".downsample.1", train)
        idt = _bn(P, pre + ".downsample.1", idt, train)
    return F.relu(y + idt)


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


# ----------------------------------------------------------------------------- map stack
def map_encoder(P, ego_map, train, pre="net.map_encoder.cnn"):
    x = _conv(P, pre + ".0", ego_map, 2, 3)
    _bump(P, pre + ".1", train)
    x = F.relu(_bn(P, pre + ".1", x, train))
    x = _conv(P, pre + ".3", x, 2, 1)
    _bump(P, pre + ".4", train)
    x = F.relu(_bn(P, pre + ".4", x, train))
    x = _conv(P, pre + ".6", x, 1, 1)
    _bump(P, pre + ".7", train)
    return F.relu(_bn(P, pre + ".7", x, train))


def map_decoder(P, x, train, pre="net.map_decoder"):
    xo = _convrelu(P, pre + ".conv_original_size0", x, 1, train)
    xo = _convrelu(P, pre + ".conv_original_size1", xo, 1, train)
    l0 = _conv(P, pre + ".base_model.conv1", x, 2, 3)
    _bump(P, pre + ".base_model.bn1", train)
    l0 = F.relu(_bn(P, pre + ".base_model.bn1", l0, train))
    l1 = F.max_pool2d(l0, 3, 2, 1)
    l1 = _basic_block(P, pre + ".base_model.layer1.0", l1, train)
    l1 = _basic_block(P, pre + ".base_model.layer1.1", l1, train)
    l1 = _convrelu(P, pre + ".layer1_1x1", l1, 0, train)
    u = _up2(l1)
    l0 = _convrelu(P, pre + ".layer0_1x1", l0, 0, train)
    u = _convrelu(P, pre + ".conv_up0", torch.cat([u, l0], 1), 1, train)
    u = _up2(u)
    return _convrelu(P, pre + ".conv_original_size2", torch.cat([u, xo], 1), 1, train)


def map_classifier(P, x, train, pre="net.map_classfier"):
    y = F.conv_transpose2d(x, P[pre + ".0.weight"], None, stride=2, padding=1)
    _bump(P, pre + ".1", train)
    y = F.relu(_bn(P, pre + ".1", y, train))
    y = _conv(P, pre + ".3", y, 1, 1)
    _bump(P, pre + ".4", train)
    y = F.relu(_bn(P, pre + ".4", y, train))
    return _conv(P, pre + ".6", y, 1, 0)


def map_stack(P, ego_map, train):
    """ego_map [B,C,E,E] -> (map_embedding [B,256,S*S], pred_sem_map [B,27,2S,2S]); mg_map_policy.py:189-207."""
    enc = map_encoder(P, ego_map, train)
    enc_proj = F.relu(_conv(P, "net.map_encoded_linear.0", enc, 1, 1))
    dec = map_decoder(P, enc, train)
    sem = map_classifier(P, dec, train)
    cls_proj = F.relu(_conv(P, "net.map_classified_linear.0", F.avg_pool2d(sem, 2, 2), 1, 1))
    emb = F.relu(_conv(P, "net.map_cated_linear.0", torch.cat([enc_proj, cls_proj], 1), 1, 1))
    return emb.flatten(2), sem


# ----------------------------------------------------------------------------- sequence models
def instruction_encoder(P, instruction, pre="net.instruction_encoder"):
    """Embedding -> packed bi-LSTM -> [B,256,Lmax], pad mask [B,Lmax] (instruction_encoder.py:75-93)."""
    tok = instruction.long()
    lengths = (tok != 0).long().sum(1)
    emb = F.embedding(tok, P[pre + ".embedding_layer.weight"])
    packed = torch.nn.utils.rnn.pack_padded_sequence(emb, lengths.cpu(), batch_first=True, enforce_sorted=False)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
             "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    lstm = torch.nn.LSTM(emb.shape[-1], P[pre + ".encoder_rnn.weight_hh_l0"].shape[1], bidirectional=True)
    out, _ = torch.func.functional_call(lstm, {n: P[pre + ".encoder_rnn." + n] for n in names}, (packed,))
    hid = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True)[0].permute(0, 2, 1)
    return hid, (hid == 0.0).all(dim=1)


def masked_gru(P, pre, x, h, masks):
    """habitat-lab v0.1.5 RNNStateEncoder restated: h_{t-1} * mask_t before every step
    (equivalent to its split-at-zeros sequence form).  x [B,in], h [1,N,H], masks [B,1]."""
    n = h.size(1)
    t = x.size(0) // n
    gru = torch.nn.GRU(x.size(1), h.size(2))
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    W = {k: P[pre + ".rnn." + k] for k in names}
    xs = x.view(t, n, -1)
    ms = masks.view(t, n, 1)
    outs = []
    hh = h
    for i in range(t):
        y, hh = torch.func.functional_call(gru, W, (xs[i:i + 1], hh * ms[i].unsqueeze(0)))
        outs.append(y)
    return torch.cat(outs, 0).view(t * n, -1), hh


def attn(q, k, v, mask=None, scale=1.0 / 16):
    """MGMapNet._attn (mg_map_policy.py:173-178)."""
    logits = torch.einsum("nc,nci->ni", q, k)
    if mask is not None:
        logits = logits - mask.float() * 1e8
    a = F.softmax(logits * scale, dim=1)
    return torch.einsum("ni,nci->nc", a, v), a


# ----------------------------------------------------------------------------- frozen RGB encoder
def resnet_unet(P, rgb, pre="net.rgb_encoder.base_model"):
    """ResNetUNet.forward, eval-mode BN (unet_encoder.py:68-111). rgb [B,H,W,3] raw 0..255."""
    x = rgb.permute(0, 3, 1, 2)
    rb = pre + ".base_model"
    xo = _convrelu(P, pre + ".conv_original_size0", x, 1, False)
    xo = _convrelu(P, pre + ".conv_original_size1", xo, 1, False)
    l0 = F.relu(_bn(P, rb + ".bn1", _conv(P, rb + ".conv1", x, 2, 3), False))
    l1 = F.max_pool2d(l0, 3, 2, 1)
    l1 = _basic_block(P, rb + ".layer1.1", _basic_block(P, rb + ".layer1.0", l1, False), False)
    l2 = _basic_block(P, rb + ".layer2.1", _basic_block(P, rb + ".layer2.0", l1, False, 2), False)
    l3 = _basic_block(P, rb + ".layer3.1", _basic_block(P, rb + ".layer3.0", l2, False, 2), False)
    l4 = _basic_block(P, rb + ".layer4.1", _basic_block(P, rb + ".layer4.0", l3, False, 2), False)
    l4 = _convrelu(P, pre + ".layer4_1x1", l4, 0, False)
    u = _up2(l4)
    u = _convrelu(P, pre + ".conv_up3", torch.cat([u, _convrelu(P, pre + ".layer3_1x1", l3, 0, False)], 1), 1, False)
    u = _up2(u)
    u = _convrelu(P, pre + ".conv_up2", torch.cat([u, _convrelu(P, pre + ".layer2_1x1", l2, 0, False)], 1), 1, False)
    u = _up2(u)
    u = _convrelu(P, pre + ".conv_up1", torch.cat([u, _convrelu(P, pre + ".layer1_1x1", l1, 0, False)], 1), 1, False)
    u = _up2(u)
    u = _convrelu(P, pre + ".conv_up0", torch.cat([u, _convrelu(P, pre + ".layer0_1x1", l0, 0, False)], 1), 1, False)
    u = _up2(u)
    proj = _convrelu(P, pre + ".conv_original_size2", torch.cat([u, xo], 1), 1, False)
    return l4, proj


# ----------------------------------------------------------------------------- frozen depth encoder (third-party)
def ddppo_resnet50(P, depth, pre="net.depth_encoder.visual_encoder", ngroups=16):
    """habitat-lab v0.1.5 ResNetEncoder(depth) with the resnet50 GroupNorm backbone, as instantiated at
    resnet_encoders.py:25-32 (baseplanes 32, ngroups 16).  Third-party: restated from its published sources
    (rl/ddppo/policy/resnet.py, resnet_policy.py) — parity UNPINNED.  depth [B,H,W,1] -> [B,128,H/64,W/64]."""
    def gn(x, k, groups):
        return F.group_norm(x, groups, P[k + ".weight"], P[k + ".bias"], 1e-5)

    x = F.avg_pool2d(depth.permute(0, 3, 1, 2), 2)
    bb = pre + ".backbone"
    x = F.relu(gn(F.conv2d(x, P[bb + ".conv1.0.weight"], None, 2, 3), bb + ".conv1.1", ngroups))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nblk in zip((1, 2, 3, 4), (3, 4, 6, 3)):
        for b in range(nblk):
            k = f"{bb}.layer{li}.{b}"
            stride = 2 if (b == 0 and li > 1) else 1
            y = F.relu(gn(F.conv2d(x, P[k + ".convs.0.weight"]), k + ".convs.1", ngroups))
            y = F.relu(gn(F.conv2d(y, P[k + ".convs.3.weight"], None, stride, 1), k + ".convs.4", ngroups))
            y = gn(F.conv2d(y, P[k + ".convs.6.weight"]), k + ".convs.7", ngroups)
            idt = x
            if (k + ".downsample.0.weight") in P:
                idt = gn(F.conv2d(x, P[k + ".downsample.0.weight"], None, stride), k + ".downsample.1", ngroups)
            x = F.relu(y + idt)
    x = F.conv2d(x, P[pre + ".compression.0.weight"], None, 1, 1)
    return F.relu(gn(x, pre + ".compression.1", 1))


# ----------------------------------------------------------------------------- the network
class PolicyRef:
    """Holds P (+ mapper state, AuxLosses-like registry) and restates BasePolicy's methods."""

    def __init__(self, P, num_proc=1, E=100, C=64, G=240, alphas=(0.1, 1.0, 1.0), tau=0.07):
        self.P = P
        self.mapper = bev_ref.MapperRef(num_proc, G, E, C)
        self.train_mode = True
        self.alpha_pred, self.alpha_con, self.alpha_prog = alphas
        self.tau = tau
        self.aux_active = False
        self.losses = {}
        self.prog = None
        self.att_map_t_m = None

    # -- MGMapNet.forward ------------------------------------------------------
    def net(self, obs, h, masks):
        P = self.P
        instr, text_mask = instruction_encoder(P, obs["instruction"])
        if "rgb_features" in obs:
            rgb_emb, proj = obs["rgb_features"], None
        else:
            rgb_emb, proj = resnet_unet(P, obs["rgb"])
        dx = obs["depth_features"] if "depth_features" in obs else ddppo_resnet50(P, obs["depth"])   # resnet_encoders.py:79-82
        b = dx.size(0)
        se = P["net.depth_encoder.spatial_embeddings.weight"]
        depth_emb = torch.cat([dx, se.view(1, -1, dx.size(2), dx.size(3)).expand(b, se.size(1), dx.size(2), dx.size(3))], 1)
        if "rgb_ego_map" not in obs:
            obs["rgb_ego_map"] = self.mapper.step(proj, obs["depth"], obs["gps"], obs["compass"], masks)
        emb, sem = map_stack(P, obs["rgb_ego_map"], self.train_mode)
        rgb_in = F.relu(F.linear(rgb_emb.flatten(2).mean(-1), P["net.rgb_linear.2.weight"], P["net.rgb_linear.2.bias"]))
        depth_in = F.relu(F.linear(depth_emb.flatten(1), P["net.depth_linear.1.weight"], P["net.depth_linear.1.bias"]))
        map_in = F.relu(F.linear(emb.mean(-1), P["net.map_linear.2.weight"], P["net.map_linear.2.bias"]))
        state, h0 = masked_gru(P, "net.state_encoder", torch.cat([rgb_in, depth_in, map_in], 1), h[0:1], masks)
        q = F.linear(state, P["net.state_text_q_layer.weight"], P["net.state_text_q_layer.bias"])
        k = F.conv1d(instr, P["net.state_text_k_layer.weight"], P["net.state_text_k_layer.bias"])
        text, _ = attn(q, k, instr, text_mask)
        q2 = F.linear(text, P["net.text_map_q_layer.weight"], P["net.text_map_q_layer.bias"])
        k2 = F.conv1d(emb, P["net.text_map_k_layer.weight"], P["net.text_map_k_layer.bias"])
        mp, self.att_map_t_m = attn(q2, k2, emb, None)
        x = torch.cat([state, text, mp], 1)
        x = F.relu(F.linear(x, P["net.second_state_compress.0.weight"], P["net.second_state_compress.0.bias"]))
        x, h1 = masked_gru(P, "net.second_state_encoder", x, h[1:2], masks)
        return x, torch.cat([h0, h1], 0), sem

    # -- BasePolicy.aux_prediction --------------------------------------------
    def aux_prediction(self, feats, obs, sem):
        P = self.P
        self.prog = torch.tanh(F.linear(feats, P["prog_pred.weight"], P["prog_pred.bias"]))
        if not self.aux_active:
            return
        target = F.interpolate(obs["gt_semantic_map"].unsqueeze(1), size=(48, 48)).squeeze(1).long()
        self.losses["prediction_monitor"] = (F.cross_entropy(sem, target, reduction="none").mean([1, 2]), self.alpha_pred)
        S = int(math.isqrt(self.att_map_t_m.shape[1]))
        dm = obs["gt_path"] if "gt_path" in obs else obs["waypoint_distribution"]
        tgt = (dm.max() - dm) / (dm.max() - dm.min())
        tgt = F.interpolate(tgt.unsqueeze(1), size=[S, S], mode="area").squeeze(1)
        tgt = F.softmax(tgt.reshape(tgt.shape[0], -1) / self.tau, dim=1)
        kl = F.kl_div(torch.log(self.att_map_t_m), tgt, reduction="none").mean(-1)
        self.losses["contrastive_monitor"] = (kl, self.alpha_con)
        self.losses["progress_monitor"] = (F.mse_loss(self.prog, obs["progress"], reduction="none").mean(-1), self.alpha_prog)

    def reduce(self, mask):
        total = 0.0
        for v, a in self.losses.values():
            total = total + a * torch.masked_select(v, mask).mean()
        return total

    # -- BasePolicy.forward / act / update_map ---------------------------------
    def forward(self, obs, h, prev_actions, masks, weights):
        P = self.P
        self.losses = {}
        feats, h, sem = self.net(obs, h, masks)
        pred = F.linear(feats, P["action_distribution.fc_mean.weight"], P["action_distribution.fc_mean.bias"])
        self.aux_prediction(feats, obs, sem)
        aux = self.reduce((weights > 0).view(-1)) if self.aux_active else 0.0
        return pred, aux, h, sem

    def act(self, obs, h, prev_actions, masks):
        """deterministic=True branch: action = mean (distributions.py:29)."""
        P = self.P
        feats, h, sem = self.net(obs, h, masks)
        self.aux_prediction(feats, obs, sem)
        mean = F.linear(feats, P["action_distribution.fc_mean.weight"], P["action_distribution.fc_mean.bias"])
        logstd = torch.zeros_like(mean) + P["action_distribution.logstd._bias"].t().view(1, -1)
        value = F.linear(feats, P["critic.fc.weight"], P["critic.fc.bias"])
        logp = torch.distributions.Normal(mean, logstd.exp()).log_prob(mean).sum(-1)
        return value, mean, logp, h

    def update_map(self, obs, masks):
        _, proj = resnet_unet(self.P, obs["rgb"])
        obs["rgb_ego_map"] = self.mapper.step(proj, obs["depth"], obs["gps"], obs["compass"], masks)


def dagger_loss(pred, aux_loss, waypoint, weights):
    """dagger_trainer.py:526-533."""
    T, N = weights.shape
    logits = torch.tanh(pred).view(T, N, -1)
    al = F.mse_loss(logits, waypoint[:, :2].view(T, N, -1), reduction="none").sum(2)
    al = ((weights * al).sum(0) / weights.sum(0)).mean()
    return al + aux_loss, al

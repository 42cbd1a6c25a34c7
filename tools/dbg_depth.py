import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from wsmgmap import ops
from wsmgmap.models.encoders.ddppo_resnet import ResNetEncoder
torch.manual_seed(0)
enc = ResNetEncoder().cuda().eval()
for m in enc.modules():
    if isinstance(m, torch.nn.GroupNorm):
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
depth = torch.rand(2, 256, 256, 1, device="cuda")
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
with torch.no_grad():
    x = F.avg_pool2d(depth.permute(0, 3, 1, 2), 2)
    bb = enc.backbone
    # stage by stage
    xe = ops.to_nhwc(x.contiguous(), 32, dtype=torch.bfloat16)
    y_ref = bb.conv1[0](x)
    y_e = ops.conv2d_infer_bf16(xe, enc._w(bb.conv1[0], 32), None, 2, 3, False)
    print("conv1 conv", rel(y_e.permute(0, 3, 1, 2), y_ref))
    g_ref = F.relu(bb.conv1[1](y_ref))
    g_e = ops.group_norm_nhwc(y_e, bb.conv1[1].weight, bb.conv1[1].bias, 16, 1e-5, True)
    print("conv1 gn", rel(g_e.permute(0, 3, 1, 2), g_ref))
    p_ref = bb.maxpool(g_ref); p_e = ops.maxpool3x3s2(g_e)
    print("maxpool", rel(p_e.permute(0, 3, 1, 2), p_ref))
    xr, xe = p_ref, p_e
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(bb, f"layer{li}")):
            xr2 = blk(xr)
            idt = xe if blk.downsample is None else enc._conv_gn(xe, blk.downsample[0], blk.downsample[1], False)
            if blk.downsample is not None:
                print("   ds", rel(idt.permute(0, 3, 1, 2), blk.downsample(xr)))
            y = enc._conv_gn(xe, blk.convs[0], blk.convs[1], True)
            print("   c0", rel(y.permute(0, 3, 1, 2), blk.convs[:3](xr)))
            y2 = enc._conv_gn(y, blk.convs[3], blk.convs[4], True)
            print("   c3", rel(y2.permute(0, 3, 1, 2), blk.convs[:6](xr)))
            xe2 = enc._conv_gn(y2, blk.convs[6], blk.convs[7], True, residual=idt)
            print(f"layer{li}.{bi}", rel(xe2.permute(0, 3, 1, 2), xr2), tuple(xr2.shape))
            xr, xe = xr2, xe2

"""CLIP RN50 visual tower (ModifiedResNet + AttentionPool2d), fp32 torch-CPU (test infrastructure).

Restates clip==1.0 ``clip/model.py`` ``ModifiedResNet.forward`` / ``Bottleneck.forward`` /
``AttentionPool2d.forward`` as reached from lossyless/architectures.py:367-371 (``clip.load("RN50")``);
the package is not vendored in /root/reference, the recipe is the published model code:
3-convolution stem with BatchNorm + ReLU and a 2x2 average pool; bottlenecks [3,4,6,3] whose strided
convolutions are replaced by an average pool after conv2 (and in front of the 1x1 downsample
convolution); attention pooling with the mean token as the single query, 32 heads, output 1024.
Parity unpinned (no reference-held vectors).  ``fp16_storage=True`` rounds every stored
activation to fp16 where the HIP tower stores fp16 (used to attribute the tower's error, as for the ViT).
"""
import torch
import torch.nn.functional as F

BLOCKS = (3, 4, 6, 3)


def rn50_forward(sd, x, fp16_storage=False, weights_rounded_to_fp16=True):
    """sd: OpenAI-layout state-dict (``visual.`` stripped); x [B,3,224,224] fp32 -> z [B,1024] fp32."""
    r = (lambda t: t.half().float()) if fp16_storage else (lambda t: t)
    eps = 1e-5

    def conv_bn(t, conv, bn, stride=1, pad=0, relu=True):
        w = sd[conv + ".weight"].float()
        scale = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + eps)
        wf = w * scale.view(-1, 1, 1, 1)
        bf = sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * scale
        if weights_rounded_to_fp16:         # what the device blob holds (BN folded in fp32, then rounded)
            wf = wf.half().float()
        y = F.conv2d(t, wf, bf, stride=stride, padding=pad)
        return F.relu(y) if relu else y

    x = r(x.float())
    x = r(conv_bn(x, "conv1", "bn1", stride=2, pad=1))
    x = r(conv_bn(x, "conv2", "bn2", pad=1))
    x = r(conv_bn(x, "conv3", "bn3", pad=1))
    x = r(F.avg_pool2d(x, 2))
    for s, nb in enumerate(BLOCKS):
        for b in range(nb):
            p = f"layer{s + 1}.{b}."
            stride = 2 if (s > 0 and b == 0) else 1
            out = r(conv_bn(x, p + "conv1", p + "bn1"))
            out = r(conv_bn(out, p + "conv2", p + "bn2", pad=1))
            if stride > 1:
                out = r(F.avg_pool2d(out, stride))
            identity = x
            if b == 0:
                identity = r(F.avg_pool2d(x, stride)) if stride > 1 else x
                identity = r(conv_bn(identity, p + "downsample.1", p + "downsample.2", relu=False))
            x = r(F.relu(conv_bn(out, p + "conv3", p + "bn3", relu=False) + identity))
    # attention pool
    B = x.shape[0]
    t = x.flatten(2).permute(0, 2, 1)                                  # [B, 49, 2048]
    h16 = (lambda w: w.half().float()) if weights_rounded_to_fp16 else (lambda w: w.float())
    pos = h16(sd["attnpool.positional_embedding"])
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + pos[None]
    t = r(t)

    def lin(v, n):
        return v @ h16(sd[f"attnpool.{n}.weight"]).t() + h16(sd[f"attnpool.{n}.bias"])

    q = r(lin(t[:, :1], "q_proj")) * 0.125
    k, v = r(lin(t, "k_proj")), r(lin(t, "v_proj"))
    q = q.view(B, 1, 32, 64).transpose(1, 2)
    k = k.view(B, 50, 32, 64).transpose(1, 2)
    v = v.view(B, 50, 32, 64).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = r((att @ v).transpose(1, 2).reshape(B, 2048))
    return lin(o, "c_proj")

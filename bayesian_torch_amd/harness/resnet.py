"""Deterministic ResNet skeletons that get fed to ``dnn_to_bnn``.

torchvision is not installed in this image, and the reference's README
(README.md:98-100) converts ``torchvision.models.resnet18()``.  This file is a
plain-PyTorch stand-in with the torchvision topology and *module names*
(``conv1, bn1, layer1.0.conv1, ..., layer4.1.bn2, fc``): 7x7/s2 stem, 3x3/s2
max-pool, four stages of residual blocks, adaptive average pool, one Linear.

It holds no Bayesian code at all -- the Bayesian layers are swapped in by
``bayesian_torch_amd.models.dnn_to_bnn.dnn_to_bnn`` exactly as the reference's
converter does for a user model (reference ``models/dnn_to_bnn.py:127-154``).

``width`` scales the channel counts (64 -> the standard model; 8 -> the tiny
model used for fully-stored golden fixtures).
"""
import torch
import torch.nn as nn


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride, shortcut):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = shortcut

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        if getattr(self, "fused", False):          # fuse_inference(): bn/relu/add live in the conv kernels' output stage
            return self.conv2(self.conv1(x), residual=skip)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + skip)


class _Bottle(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, shortcut):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        # stride sits on the 3x3, as in reference models/deterministic/resnet_large.py:72-77
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = shortcut

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        if getattr(self, "fused", False):
            return self.conv3(self.conv2(self.conv1(x)), residual=skip)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


class ResNet(nn.Module):
    def __init__(self, block, depths, num_classes=10, width=64, in_ch=3):
        super().__init__()
        self._cin = width
        self.conv1 = nn.Conv2d(in_ch, width, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, width, depths[0], 1)
        self.layer2 = self._stage(block, width * 2, depths[1], 2)
        self.layer3 = self._stage(block, width * 4, depths[2], 2)
        self.layer4 = self._stage(block, width * 8, depths[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(width * 8 * block.expansion, num_classes)

    def _stage(self, block, planes, n, stride):
        cout = planes * block.expansion
        shortcut = None
        if stride != 1 or self._cin != cout:
            shortcut = nn.Sequential(nn.Conv2d(self._cin, cout, 1, stride, bias=False),
                                     nn.BatchNorm2d(cout))
        mods = [block(self._cin, planes, stride, shortcut)]
        self._cin = cout
        mods += [block(cout, planes, 1, None) for _ in range(n - 1)]
        return nn.Sequential(*mods)

    def forward(self, x):
        if getattr(self, "fused", False):
            x = self.conv1(x)  # bn1 / relu / maxpool live in its output stage
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        if x.shape[-2:] != (1, 1):     # (CIFAR-sized inputs end on 1x1 maps: the mean of one element is that element -- no launch)
            x = self.avgpool(x)
        return self.fc(torch.flatten(x, 1))


def resnet18(num_classes=10, width=64):
    return ResNet(_Basic, [2, 2, 2, 2], num_classes, width)


def resnet50(num_classes=1000, width=64):
    return ResNet(_Bottle, [3, 4, 6, 3], num_classes, width)


def mlp(sizes=(3072, 512, 10)):
    """cfg2 of BASELINE.json: Linear -> ReLU -> Linear (...)."""
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i + 2 < len(sizes):
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


def bayes_layers(model):
    """Bayesian layers of a converted model, in ``named_modules`` order."""
    return [(n, m) for n, m in model.named_modules()
            if hasattr(m, "mu_kernel") or hasattr(m, "mu_weight")]


def fill_bayes_params(model, seed, mu_std=0.1, rho_mean=-3.0, rho_std=0.1):
    """Overwrite every (mu, rho) with a recipe that depends only on (seed, layer
    index, shape) -- so a reference-converted model and a model converted by this
    package get bit-identical parameters without sharing construction-time RNG.
    Same distribution as the layers' own init (reference linear_variational.py:137-138)."""
    with torch.no_grad():
        for i, (_, m) in enumerate(bayes_layers(model)):
            g = torch.Generator().manual_seed(seed * 1000 + i)
            for nm in ("mu_kernel", "rho_kernel", "mu_weight", "rho_weight", "mu_bias", "rho_bias"):
                p = getattr(m, nm, None)
                if p is None:
                    continue
                v = torch.randn(p.shape, generator=g)
                v = v * (mu_std if nm.startswith("mu") else rho_std) + (0.0 if nm.startswith("mu") else rho_mean)
                p.copy_(v.to(p.device))


def fuse_inference(model):
    """Fold every BatchNorm / ReLU / residual add of a converted ResNet into the Bayesian convs' output stage
    (bayesian_torch_amd.fuse).  The model must be converted (dnn_to_bnn), on its device and in eval()."""
    from ..fuse import fold_pair, fold_maxpool
    fold_pair(model.conv1, model.bn1, relu=True)
    fold_maxpool(model.conv1, model.maxpool)
    for stage in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in stage:
            last = 3 if isinstance(blk, _Bottle) else 2
            for i in range(1, last + 1):
                fold_pair(getattr(blk, f"conv{i}"), getattr(blk, f"bn{i}"), relu=True)   # the last ReLU runs after the fused add
            if blk.downsample is not None:
                fold_pair(blk.downsample[0], blk.downsample[1], relu=False)
                blk.downsample[1] = nn.Identity()
            blk.fused = True
    model.fused = True
    return model

"""VGG16 conv1_1 .. conv5_3 forward in PyTorch-ROCm (fp32).

This is the one place torch computes anything: north_star keeps the backbone on
PyTorch-ROCm and the hand-written HIP path starts at its output.  Architecture per
models/Pascal/VGG16/az-net/test.prototxt:16-384: thirteen 3x3 pad-1 convs + ReLU, four
2x2/2 max-pools in Caffe's ceil mode (600x1000 -> 38x63), no pool5.  torchvision is not
available offline, so the stack is spelled out here; weights are seeded He-normal unless
a dict of Caffe-layout arrays is supplied.
"""
import numpy as np
import torch
import torch.nn.functional as F

# (name, out_channels) with 'P' = max-pool
VGG16_CONV = [("conv1_1", 64), ("conv1_2", 64), "P", ("conv2_1", 128), ("conv2_2", 128), "P",
              ("conv3_1", 256), ("conv3_2", 256), ("conv3_3", 256), "P",
              ("conv4_1", 512), ("conv4_2", 512), ("conv4_3", 512), "P",
              ("conv5_1", 512), ("conv5_2", 512), ("conv5_3", 512)]


class VGG16Conv5(object):
    def __init__(self, device="cuda:0", seed=4321, weights=None, width_div=1, channels_last_out=False,
                 channels_last_compute=None):
        """width_div > 1 shrinks every layer's channel count (fast tests); 1 = real VGG16.
        channels_last_out: return conv5_3 in torch.channels_last memory ([H][W][C], the layout RoIPool reads):
        the HIP context then borrows it without its own transpose.
        channels_last_compute (default off): weights and activations in channels_last memory.  With
        torch.backends.cudnn.benchmark = True set BEFORE the first forward, MIOpen then finds fp32 convolutions that run
        ~12 % faster on MI355X (4.09 -> 3.61 ms for a 600x1000 image); without the benchmark search the same layout
        falls on a slower default (6.0 ms), which is why it is opt-in.  Same arithmetic type either way."""
        self.device = torch.device(device)
        self.channels_last_out = bool(channels_last_out)
        self.cl_compute = bool(channels_last_compute) and self.device.type == "cuda"
        self.fused_epilogue = True          # (False: PyTorch's own bias / ReLU / pool launches -- tests compare the two)
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.layers = []
        cin = 3
        for item in VGG16_CONV:
            if item == "P":
                self.layers.append(None)
                continue
            name, cout = item
            cout = max(4, cout // width_div)
            if weights is not None and name in weights:
                w = torch.from_numpy(np.ascontiguousarray(weights[name][0], dtype=np.float32))
                b = torch.from_numpy(np.ascontiguousarray(weights[name][1], dtype=np.float32))
            else:
                w = torch.randn(cout, cin, 3, 3, generator=g) * float(np.sqrt(2.0 / (cin * 9)))
                b = torch.zeros(cout)
            w = w.to(self.device)
            if self.cl_compute:
                w = w.contiguous(memory_format=torch.channels_last)
            self.layers.append((name, w, b.to(self.device)))
            cin = int(w.shape[0])
        self.out_channels = cin

    @torch.no_grad()
    def forward(self, blob):
        """blob: [1,3,H,W] float32 (BGR, mean-subtracted), NumPy or torch -> conv5_3 [1,C,h,w]
        contiguous fp32 tensor on the device."""
        x = torch.as_tensor(blob, dtype=torch.float32, device=self.device)
        if self.cl_compute:
            x = x.contiguous(memory_format=torch.channels_last)
        # On the GPU: what follows a convolution -- bias, ReLU, and the pooling layer where there is one --
        # is ONE pass over its output (az_bias_relu / az_bias_relu_pool, az_epilogue.hip) instead of PyTorch's two or three
        # element-wise launches; same fp32 operations, same bits.  The convolutions are PyTorch-ROCm's either way.
        fused = self.fused_epilogue and x.is_cuda and x.shape[0] == 1
        skip_pool = False
        for li, layer in enumerate(self.layers):
            if layer is None:
                if not skip_pool:
                    x = F.max_pool2d(x, kernel_size=2, stride=2, ceil_mode=True)
                skip_pool = False
                continue
            if fused and (layer[1].shape[0] % 4 == 0 or not self.cl_compute):
                y = F.conv2d(x, layer[1], None, padding=1)
                if (y.is_contiguous(memory_format=torch.channels_last) if self.cl_compute else y.is_contiguous()) \
                        and y.data_ptr() % 16 == 0 and layer[2].data_ptr() % 16 == 0 and layer[2].is_contiguous():
                    from . import ffi
                    try:
                        if li + 1 < len(self.layers) and self.layers[li + 1] is None:
                            x = ffi.bias_relu_pool(y, layer[2])
                            skip_pool = True
                        else:
                            x = ffi.bias_relu_(y, layer[2])
                        continue
                    except ffi.AzError as e:
                        # (a layout the fused kernels decline -- AZ_ERR_INVALID, nothing was written: PyTorch's own ops)
                        if e.code != ffi.AZ_ERR_INVALID:
                            raise
                        skip_pool = False
                x = F.relu_(y + layer[2].view(1, -1, 1, 1))
                continue
            x = F.relu_(F.conv2d(x, layer[1], layer[2], padding=1))
        if self.channels_last_out:
            return x.contiguous(memory_format=torch.channels_last)
        return x.contiguous()

    __call__ = forward

    @torch.no_grad()
    def normalize_output(self, blob):
        """Synthetic (random-init) weights only: rescale conv5_3's filters so the map has
        unit RMS on `blob` -- stands in for what training would have done, and keeps the
        synthetic head's scores and box deltas in a sane range."""
        rms = float(self.forward(blob).pow(2).mean().sqrt())
        name, w, b = self.layers[-1]
        w = w / rms
        if self.cl_compute:
            w = w.contiguous(memory_format=torch.channels_last)
        self.layers[-1] = (name, w, b / rms)
        return rms

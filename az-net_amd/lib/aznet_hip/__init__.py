"""aznet_hip: MI355X-native backend of AZ-Net's proposal search.

`ffi`      ctypes binding of libaznet_hip.so (the C ABI in include/aznet_hip.h)
`net`      HipAZNet: the object that stands where the reference's caffe.Net pair stood
`backbone` VGG16 conv1_1..conv5_3 forward in PyTorch-ROCm (plumbing, not the product)
`dist`     one-rank-per-GPU image sharding and the RCCL gather of proposals
`synth`    seeded synthetic weights / images (no datasets or .caffemodel offline)
"""

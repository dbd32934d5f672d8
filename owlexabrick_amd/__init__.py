"""owlexabrick_amd — MI355X-native ExaBrick DVR / implicit-iso renderer.

Host-side Python here is plumbing around the C-ABI HIP module
(owlexabrick_amd/csrc -> libexa_hip.so, declared in include/exa_hip.h).
"""

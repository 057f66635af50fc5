# `make san` (fthmc_amd/csrc/Makefile): the recipe of the host-side sanitizer build.  Listed in .gpurunignore: the GPU pool refuses
# snapshots whose build files ask for sanitizers.
include Makefile

# Sanitizer build of the HOST side (never run on a GPU box): the host pass of every source with AddressSanitizer + UBSan
# (-Xarch_host: the device pass is what the product builds), launches / copies / memsets as succeeding no-ops (-DFT_DRYRUN,
# common.h), and the operator library against it.  tests/test_sanitizer.py loads them under the ASan runtime and takes every
# entry point through its argument checks, workspace carving and launch sequencing.
SANDIR = /tmp/fthmc_san
SANOUT = ../libfthmc_hip_san.so
SANTORCH = ../libfthmc_torch_san.so
SANFLAGS = -O1 -g -fno-omit-frame-pointer -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize-recover=undefined -DFT_DRYRUN -Wno-unused
# the operator library is built by clang++ here (ONE sanitizer runtime in the process): -fclang-abi-compat=17 keeps the mangling of
# PyTorch's enable_if templates the one g++ gave libtorch; -asan-globals=0: libstdc++'s header string constants exist in the
# uninstrumented libtorch too, instrumented twins of them trip the runtime's global registration
SANRT := $(shell ls $(ROCM)/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so 2>/dev/null | head -1)
san_build:
	mkdir -p $(SANDIR)
	for f in $(SRCS:.hip=); do $(HIPCC) -std=c++17 -fPIC --offload-arch=$(ARCH) $(SANFLAGS) -DFTHMC_SRC_SHA=\"$(SRC_SHA)\" -c $$f.hip -o $(SANDIR)/$$f.o & done; wait
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -Xarch_host -fsanitize=address,undefined -shared-libsan -o $(SANOUT) $(SRCS:%.hip=$(SANDIR)/%.o)
ifneq ($(TORCHDIR),)
	$(ROCM)/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -fclang-abi-compat=17 -mllvm -asan-globals=0 \
	  -D_GLIBCXX_USE_CXX11_ABI=$(TORCHABI) -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 \
	  -I$(TORCHDIR)/include -I$(TORCHDIR)/include/torch/csrc/api/include -I$(ROCM)/include torch_library.cpp -o $(SANTORCH) \
	  -L.. -lfthmc_hip_san -L$(TORCHDIR)/lib -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(TORCHDIR)/lib
endif
	@echo "sanitizer runtime to preload: $(SANRT)"


# Builds the product library (HIP, gfx950) and the test oracle (C).
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -fPIC -shared -fwrapv -ffp-contract=off -std=c++17 -Wall -Wextra
SRC      = mrnet_amd/csrc
LIB      = mrnet_amd/lib/libmrpdr.so

HOSTTEST = tests/host/test_packet_wave
CTEST    = tests/host/test_c_abi
PLUGTEST = tests/host/test_plugin
RCCLTEST = tests/host/test_rccl_combine
ROCM    ?= /opt/rocm

all: $(LIB) oracle $(HOSTTEST) $(CTEST) plugin-check $(PLUGTEST) $(RCCLTEST)

$(LIB): $(SRC)/mrpdr.hip $(wildcard $(SRC)/*.h) include/mrpdr.h
	mkdir -p mrnet_amd/lib
	$(HIPCC) $(HIPFLAGS) -o $@ $(SRC)/mrpdr.hip

oracle:
	$(MAKE) -C oracle

# C++ host layer (mrnet_amd/host/PacketWave.h) exercised by a reference-style test program
$(HOSTTEST): tests/host/test_packet_wave.cpp mrnet_amd/host/PacketWave.h include/mrpdr.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -Wall -o $@ tests/host/test_packet_wave.cpp -Lmrnet_amd/lib -lmrpdr \
		-Wl,-rpath,'$$ORIGIN/../../mrnet_amd/lib'

# the C ABI from plain C: gcc, C99, the HIP runtime only for memory
$(CTEST): tests/host/test_c_abi.c include/mrpdr.h $(LIB)
	gcc -std=c99 -O2 -Wall -Wextra -o $@ tests/host/test_c_abi.c -Lmrnet_amd/lib -lmrpdr \
		-L$(ROCM)/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../mrnet_amd/lib' -Wl,-rpath,$(ROCM)/lib

# The plugin run without MRNet: tests/host/test_plugin.cpp implements the Packet methods it calls
$(PLUGTEST): tests/host/test_plugin.cpp integration/mrpdr_filters.C integration/buildcheck/mrnet/MRNet.h include/mrpdr.h $(LIB)
	g++ -std=c++11 -O2 -Wall -D__HIP_PLATFORM_AMD__ -Iintegration/buildcheck -Iinclude -I$(ROCM)/include -o $@ \
		tests/host/test_plugin.cpp -Lmrnet_amd/lib -lmrpdr -L$(ROCM)/lib -lamdhip64 -lpthread \
		-Wl,-rpath,'$$ORIGIN/../../mrnet_amd/lib' -Wl,-rpath,$(ROCM)/lib

# One filter call over a wave held by several GPUs, from C++: RCCL for the exchange of the 16-byte nodes
$(RCCLTEST): tests/host/test_rccl_combine.cpp include/mrpdr.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -Wall -ffp-contract=off -o $@ tests/host/test_rccl_combine.cpp -Lmrnet_amd/lib -lmrpdr \
		-L$(ROCM)/lib -lrccl -lpthread -Wl,-rpath,'$$ORIGIN/../../mrnet_amd/lib' -Wl,-rpath,$(ROCM)/lib

# The MRNet filter plugin compiles (objects only: it links against a real MRNet tree, which is not
# here).  integration/buildcheck/ holds a stand-in for the few MRNet declarations it uses.
plugin-check: integration/mrpdr_filters.C integration/buildcheck/mrnet/MRNet.h include/mrpdr.h
	g++ -std=c++11 -Wall -Wextra -fPIC -fsyntax-only -D__HIP_PLATFORM_AMD__ -Iintegration/buildcheck -Iinclude \
		-I$(ROCM)/include integration/mrpdr_filters.C

# The host side of the library and the three host test programs under AddressSanitizer (host code only:
# -Xarch_host keeps the device code as it is), built with the same clang so that one ASan runtime serves all:
#   make asan-host && ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 tools/asan/test_plugin
CLANGBIN = $(ROCM)/lib/llvm/bin
asan-host:
	mkdir -p tools/asan
	$(HIPCC) --offload-arch=$(ARCH) -O1 -g -fPIC -shared -fwrapv -ffp-contract=off -std=c++17 -Xarch_host -fsanitize=address \
		-Xarch_host -fno-omit-frame-pointer -o tools/asan/libmrpdr.so $(SRC)/mrpdr.hip
	$(CLANGBIN)/clang -std=c99 -O1 -g -fsanitize=address -D__HIP_PLATFORM_AMD__ -Iinclude -I$(ROCM)/include -o tools/asan/test_c_abi \
		tests/host/test_c_abi.c -Ltools/asan -lmrpdr -L$(ROCM)/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib
	$(CLANGBIN)/clang++ -std=c++11 -O1 -g -fsanitize=address -D__HIP_PLATFORM_AMD__ -Iintegration/buildcheck -Iinclude -I$(ROCM)/include \
		-o tools/asan/test_plugin tests/host/test_plugin.cpp -Ltools/asan -lmrpdr -L$(ROCM)/lib -lamdhip64 -lpthread \
		-Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM)/lib
	$(HIPCC) -O1 -g -std=c++17 -Xarch_host -fsanitize=address -o tools/asan/test_packet_wave tests/host/test_packet_wave.cpp \
		-Ltools/asan -lmrpdr -Wl,-rpath,'$$ORIGIN'

clean:
	rm -f $(LIB) $(HOSTTEST) $(CTEST) $(PLUGTEST) $(RCCLTEST)
	rm -rf tools/asan
	$(MAKE) -C oracle clean

.PHONY: all oracle clean plugin-check asan-host

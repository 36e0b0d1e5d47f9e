# Builds the product library (HIP, gfx950) and the test oracle (C).
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -fPIC -shared -fwrapv -ffp-contract=off -std=c++17 -Wall -Wextra
SRC      = mrnet_amd/csrc
LIB      = mrnet_amd/lib/libmrpdr.so

HOSTTEST = tests/host/test_packet_wave
CTEST    = tests/host/test_c_abi
ROCM    ?= /opt/rocm

all: $(LIB) oracle $(HOSTTEST) $(CTEST)

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

clean:
	rm -f $(LIB) $(HOSTTEST) $(CTEST)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean

# Builds the product library (HIP, gfx950) and the test oracle (C).
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -fPIC -shared -fwrapv -ffp-contract=off -std=c++17 -Wall -Wextra
SRC      = mrnet_amd/csrc
LIB      = mrnet_amd/lib/libmrpdr.so

all: $(LIB) oracle

$(LIB): $(SRC)/mrpdr.hip $(SRC)/common.h $(SRC)/scan.h $(SRC)/rows.h $(SRC)/copy.h $(SRC)/filter.h $(SRC)/header.h include/mrpdr.h
	mkdir -p mrnet_amd/lib
	$(HIPCC) $(HIPFLAGS) -o $@ $(SRC)/mrpdr.hip

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean

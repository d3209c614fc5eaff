# Builds the product: librkmh_amd.so (HIP kernels + C ABI, gfx950 only) and the rkmh CLI.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH := gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $(EXTRA_HIPFLAGS)
CSRC := rkmh_amd/csrc
# instantiations of the fused kernel the ISA lint must find (it fails closed below that): k = 12, 16 (x 3 folds), 20, 21, 31 and run-time k, x 5 modes x 3 prefetch depths
MIN_TILE_KERNELS ?= 120
LIB := rkmh_amd/lib/librkmh_amd.so
OBJS := $(CSRC)/rk_kernels.o $(CSRC)/rk_classify.o $(CSRC)/rk_kmer.o $(CSRC)/rk_count.o $(CSRC)/rk_call.o $(CSRC)/rk_sort.o $(CSRC)/rk_fastq.o $(CSRC)/rk_fasta.o $(CSRC)/rk_api.o $(CSRC)/rk_parse.o $(CSRC)/rk_format.o $(CSRC)/rk_synth.o

all: $(LIB) bin/rkmh oracle

$(CSRC)/rk_kernels.o: $(CSRC)/rk_kernels.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
# The fused kernel issues its prefetches through inline asm; the ISA of THIS compile is checked for uses of a register
# that an asm-issued load may still be writing (tools/lint_async_loads.py).  A finding fails the build.
$(CSRC)/rk_classify.o: $(CSRC)/rk_classify.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp tools/lint_async_loads.py
	@mkdir -p build/isa
	cd build/isa && $(HIPCC) $(HIPFLAGS) -save-temps -c $(CURDIR)/$< -o $(CURDIR)/$@.tmp
	@cd build/isa && rm -f *.bc *.hipi *.out *.resolution.txt *.hipfb *host-x86_64*.s *.o
	python3 tools/lint_async_loads.py --min-tile-kernels $(MIN_TILE_KERNELS) build/isa/rk_classify-hip-amdgcn-amd-amdhsa-$(ARCH).s || { rm -f $@.tmp; exit 1; }
	mv $@.tmp $@
# the k-mer-space kernel: its ISA is kept next to the fused kernel's (tools/isa_blocks.py, instruction counts in DESIGN.md)
$(CSRC)/rk_kmer.o: $(CSRC)/rk_kmer.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp
	@mkdir -p build/isa_kmer
	cd build/isa_kmer && $(HIPCC) $(HIPFLAGS) -save-temps -c $(CURDIR)/$< -o $(CURDIR)/$@
	@cd build/isa_kmer && rm -f *.bc *.hipi *.out *.resolution.txt *.hipfb *host-x86_64*.s *.o
$(CSRC)/rk_count.o: $(CSRC)/rk_count.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_sort.o: $(CSRC)/rk_sort.hip $(CSRC)/rk_kernels.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_fasta.o: $(CSRC)/rk_fasta.hip $(CSRC)/rk_kernels.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_fastq.o: $(CSRC)/rk_fastq.hip $(CSRC)/rk_kernels.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_call.o: $(CSRC)/rk_call.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_api.o: $(CSRC)/rk_api.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp include/rkmh_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_parse.o: $(CSRC)/rk_parse.cpp include/rkmh_amd.h
	g++ -O3 -std=c++17 -fPIC -Wall -c $< -o $@
$(CSRC)/rk_format.o: $(CSRC)/rk_format.cpp include/rkmh_amd.h
	g++ -O3 -std=c++17 -fPIC -Wall -c $< -o $@

$(CSRC)/rk_synth.o: $(CSRC)/rk_synth.cpp include/rkmh_amd.h
	g++ -O3 -std=c++17 -fPIC -Wall -pthread -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p rkmh_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lz -lpthread -ldl

bin/rkmh: $(CSRC)/rkmh_main.cpp $(LIB) include/rkmh_amd.h
	@mkdir -p bin
	g++ -O2 -std=c++17 -Wall -o $@ $(CSRC)/rkmh_main.cpp -Irkmh_amd/csrc -Lrkmh_amd/lib -lrkmh_amd -Wl,-rpath,'$$ORIGIN/../rkmh_amd/lib'

oracle:
	$(MAKE) -s -C oracle

clean:
	$(RM) $(OBJS) $(LIB) bin/rkmh
	$(MAKE) -s -C oracle clean
.PHONY: all clean oracle

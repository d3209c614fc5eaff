# Builds the product: librkmh_amd.so (HIP kernels + C ABI, gfx950 only) and the rkmh CLI.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH := gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $(EXTRA_HIPFLAGS)
CSRC := rkmh_amd/csrc
LIB := rkmh_amd/lib/librkmh_amd.so
OBJS := $(CSRC)/rk_kernels.o $(CSRC)/rk_classify.o $(CSRC)/rk_kmer.o $(CSRC)/rk_count.o $(CSRC)/rk_call.o $(CSRC)/rk_sort.o $(CSRC)/rk_fastq.o $(CSRC)/rk_fasta.o $(CSRC)/rk_inflate.o $(CSRC)/rk_api.o $(CSRC)/rk_index.o $(CSRC)/rk_counters.o $(CSRC)/rk_frontend.o $(CSRC)/rk_gunzip.o $(CSRC)/rk_packed.o $(CSRC)/rk_pack.o $(CSRC)/rk_parse.o $(CSRC)/rk_format.o $(CSRC)/rk_synth.o $(CSRC)/rk_policy.o

API_DEPS := $(CSRC)/rk_api_internal.hpp $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp include/rkmh_amd.h

all: $(LIB) bin/rkmh oracle

$(CSRC)/rk_kernels.o: $(CSRC)/rk_kernels.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
# the hash-space kernel: its ISA is kept under build/isa (tools/isa_blocks.py)
$(CSRC)/rk_classify.o: $(CSRC)/rk_classify.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_device.hpp
	@mkdir -p build/isa
	cd build/isa && $(HIPCC) $(HIPFLAGS) -save-temps -c $(CURDIR)/$< -o $(CURDIR)/$@
	@cd build/isa && rm -f *.bc *.hipi *.out *.resolution.txt *.hipfb *host-x86_64*.s *.o
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
$(CSRC)/rk_fastq.o: $(CSRC)/rk_fastq.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_filter_rule.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_inflate.o: $(CSRC)/rk_inflate.hip $(CSRC)/rk_kernels.hpp $(CSRC)/rk_crc32.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_call.o: $(CSRC)/rk_call.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_api.o: $(CSRC)/rk_api.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_index.o: $(CSRC)/rk_index.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_counters.o: $(CSRC)/rk_counters.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_frontend.o: $(CSRC)/rk_frontend.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_gunzip.o: $(CSRC)/rk_gunzip.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_packed.o: $(CSRC)/rk_packed.hip $(API_DEPS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/rk_pack.o: $(CSRC)/rk_pack.cpp $(CSRC)/rk_filter_rule.hpp include/rkmh_amd.h
	g++ -O3 -std=c++17 -fPIC -Wall -c $< -o $@
$(CSRC)/rk_parse.o: $(CSRC)/rk_parse.cpp include/rkmh_amd.h
	g++ -O3 -std=c++17 -fPIC -Wall -c $< -o $@
$(CSRC)/rk_format.o: $(CSRC)/rk_format.cpp $(CSRC)/rk_filter_rule.hpp include/rkmh_amd.h
	g++ -O3 -std=c++17 -fPIC -Wall -c $< -o $@

$(CSRC)/rk_policy.o: $(CSRC)/rk_policy.cpp include/rkmh_amd.h
	g++ -O2 -std=c++17 -fPIC -Wall -c $< -o $@
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

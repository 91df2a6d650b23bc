# Builds the drop-in C-ABI library (HIP kernels for gfx950 + C host planner).
#   make            -> fftw3_amd/lib/libfftw3_amd.so (+ libfftw3.so.3 alias)
#   make oracle     -> oracle/liboracle.so (CPU restatement, test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
ARCH ?= gfx950
CSRC := fftw3_amd/csrc
LIBDIR := fftw3_amd/lib
CFLAGS := -O2 -fPIC -std=gnu99 -Wall -Wextra -Iinclude -I$(CSRC)
HIPFLAGS := -O3 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -std=c++17 -Wall

OBJS := $(CSRC)/api.o $(CSRC)/planner.o $(CSRC)/sharded.o $(CSRC)/slab.o $(CSRC)/hostmath.o $(CSRC)/kernels.o $(CSRC)/kernels_rr.o $(CSRC)/kernels_rr1.o $(CSRC)/kernels_rr2.o $(CSRC)/kernels_r3.o $(CSRC)/kernels_r3r.o $(CSRC)/kernels_r2cm.o $(CSRC)/kernels_blue.o $(CSRC)/kernels_r1.o $(CSRC)/kernels_sq.o $(CSRC)/kernels_r3w.o $(CSRC)/kernels_r3tw.o $(CSRC)/kernels_bluew.o

all: $(LIBDIR)/libfftw3_amd.so

$(CSRC)/%.o: $(CSRC)/%.c $(CSRC)/fa_plan.h $(CSRC)/fa_hip.h $(CSRC)/split_costs.inc $(CSRC)/split_align.inc $(CSRC)/split2_costs.inc $(CSRC)/split2_align.inc include/fftw3.h include/fftw3_amd.h
	$(CC) $(CFLAGS) -c $< -o $@

# every HIP unit depends on exactly the headers it includes (a full rebuild of the two menu units takes minutes)
HIPCOMMON := $(CSRC)/common.hpp $(CSRC)/butterflies.h $(CSRC)/fa_hip.h include/fftw3_amd.h $(CSRC)/pass1024.hpp
$(CSRC)/kernels.o: $(CSRC)/kernels.hip $(HIPCOMMON) $(CSRC)/r2r_epi.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_rr.o: $(CSRC)/kernels_rr.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3w.hpp $(CSRC)/rr_dispatch.hpp $(CSRC)/rr_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_rr1.o: $(CSRC)/kernels_rr1.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/rr_dispatch.hpp $(CSRC)/rr_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_rr2.o: $(CSRC)/kernels_rr2.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/rr_dispatch.hpp $(CSRC)/rr_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_r3.o: $(CSRC)/kernels_r3.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3g.hpp $(CSRC)/r2crows.hpp $(CSRC)/r2r_epi.hpp $(CSRC)/r3_menu.inc $(CSRC)/r3t_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/kernels_r3r.o: $(CSRC)/kernels_r3r.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3g.hpp $(CSRC)/r3r_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_r2cm.o: $(CSRC)/kernels_r2cm.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/r2crows.hpp $(CSRC)/r2r_epi.hpp $(CSRC)/r2cr_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_bluew.o: $(CSRC)/kernels_bluew.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3g.hpp $(CSRC)/pass3b.hpp $(CSRC)/bluew_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/kernels_blue.o: $(CSRC)/kernels_blue.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3g.hpp $(CSRC)/pass3b.hpp $(CSRC)/blue_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/kernels_r1.o: $(CSRC)/kernels_r1.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass1r.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/kernels_sq.o: $(CSRC)/kernels_sq.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3q.hpp
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/kernels_r3w.o: $(CSRC)/kernels_r3w.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3g.hpp $(CSRC)/r3w_menu.inc $(CSRC)/r3rw_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/kernels_r3tw.o: $(CSRC)/kernels_r3tw.hip $(HIPCOMMON) $(CSRC)/passrr.hpp $(CSRC)/pass3s.hpp $(CSRC)/pass3g.hpp $(CSRC)/r3tw_menu.inc
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libfftw3_amd.so: $(OBJS)
	mkdir -p $(LIBDIR)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -lm -lpthread -ldl
	ln -sf libfftw3_amd.so $(LIBDIR)/libfftw3.so.3
	ln -sf libfftw3_amd.so $(LIBDIR)/libfftw3.so

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIBDIR)/*.so*

.PHONY: all oracle clean

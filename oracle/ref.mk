# Builds the one piece of the reference that compiles from its own sources:
# tools/artificial/artificial.cpp (needs only tools/fromFlash/Cmdline.h).
# Sources are compiled where they lie under /root/reference; outputs go to
# oracle/_ref/ (git-ignored).  The renderer itself (OptiX/OWL/CUDA) is unbuildable here.
REF ?= /root/reference
all: _ref/exaArtificial
_ref/exaArtificial: $(REF)/tools/artificial/artificial.cpp
	mkdir -p _ref
	g++ -O2 -std=c++14 -I$(REF)/tools/fromFlash -o $@ $<

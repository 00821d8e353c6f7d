#!/bin/sh
# Builds the C / OpenMP form of the CPU oracle (test infrastructure) into oracle/_build/ (git-ignored, travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")/../.."
mkdir -p oracle/_build
gcc -O2 -fopenmp -shared -fPIC -ffp-contract=off -o oracle/_build/libbox_oracle.so oracle/c/box_oracle.c -lm
echo oracle/_build/libbox_oracle.so

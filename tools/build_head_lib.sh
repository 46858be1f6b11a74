#!/bin/bash
# Build the committed (HEAD) sources into tss-with-dprnn_amd/libA_head.so for a same-box A/B against the working tree.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
git archive HEAD tss-with-dprnn_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/tss-with-dprnn_amd/csrc" -j8 >/dev/null
cp "$tmp/tss-with-dprnn_amd/libdprnn_mi355x.so" tss-with-dprnn_amd/libA_head.so
rm -rf "$tmp"
echo built tss-with-dprnn_amd/libA_head.so

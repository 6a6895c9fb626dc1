#!/bin/bash
# dev: build armour_amd/lib/libarmour_hip_<name>.so from a copy of the CURRENT tree with extra hipcc flags for p1_reach.hip
# usage: tools/mkvariant_tree.sh <name> [flags...]
set -e
cd /root/repo
name=$1; shift
rm -rf /tmp/var_$name && mkdir -p /tmp/var_$name && cp -r armour_amd include /tmp/var_$name/ && rm -rf /tmp/var_$name/armour_amd/lib /tmp/var_$name/armour_amd/bin
bash tools/build_variant.sh $name "$@" 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|\^" | tail -3

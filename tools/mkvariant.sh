#!/bin/bash
# dev: tools/mkvariant.sh <name> [flags...]  -- copies the source tree to /tmp/var_<name> and builds libarmour_hip_<name>.so from it (tools/build_variant.sh)
name=$1; shift
rm -rf /tmp/var_$name; mkdir -p /tmp/var_$name/armour_amd; cp -r /root/repo/armour_amd/csrc /tmp/var_$name/armour_amd/; cp -r /root/repo/include /tmp/var_$name/
exec /root/repo/tools/build_variant.sh $name "$@"

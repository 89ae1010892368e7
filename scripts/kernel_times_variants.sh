# scripts/kernel_times.sh for the default library and every build under lib/variants/.  usage: bash scripts/kernel_times_variants.sh <tag> <grep pattern>
for lib in "" $(ls neural-color-transfer_amd/lib/variants/*.so 2>/dev/null); do
  name=default; unset NCT_LIB; [ -n "$lib" ] && name=$(basename $lib .so) && export NCT_LIB=$PWD/$lib
  echo "== $name"; bash scripts/kernel_times.sh $1/$name "$2"
done

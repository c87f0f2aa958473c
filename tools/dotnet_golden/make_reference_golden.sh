#!/bin/bash
# Turns the encoder's parity from "unpinned" into pinned: runs the REAL reference (the C# sources under $SHARPZIPLIB_SRC, default
# /root/reference/src/ICSharpCode.SharpZipLib) over the golden inputs and writes tests/golden/reference_golden.json, which
# tests/test_golden.py then compares with oracle/ case by case (it skips, loudly, while the file is absent).
#   needs: a .NET SDK >= 6 (`dotnet`), python3 + numpy, gcc (for the corpus generator); ~2 GiB of scratch for the inputs with --headline.
#   usage: tools/dotnet_golden/make_reference_golden.sh [--headline | --headline-all]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
command -v dotnet >/dev/null || { echo "no dotnet SDK on this box: the reference cannot run here (DESIGN.md §2)"; exit 3; }
SRC="${SHARPZIPLIB_SRC:-/root/reference/src/ICSharpCode.SharpZipLib}"
[ -f "$SRC/ICSharpCode.SharpZipLib.csproj" ] || { echo "SHARPZIPLIB_SRC=$SRC holds no ICSharpCode.SharpZipLib.csproj"; exit 3; }
WORK="${REFGOLDEN_WORK:-$(mktemp -d)}"
python3 "$HERE/dump_inputs.py" "$WORK/inputs" "$@"
# the project is built in a scratch copy so that nothing is written next to the read-only reference checkout
mkdir -p "$WORK/proj" && cp "$HERE/RefGolden.csproj" "$HERE/Program.cs" "$WORK/proj/"
DOTNET_gcServer=0 dotnet run -c Release --project "$WORK/proj/RefGolden.csproj" -p:SharpZipLibSrc="$SRC" -- "$WORK/inputs" "$ROOT/tests/golden/reference_golden.json"
python3 -m pytest "$ROOT/tests/test_golden.py" -q -k reference

# --- the shim compiles into the reference's assembly: copy the reference project to scratch, swap the four
# files for sharpziplib_amd/dotnet/*.cs and build it.  tests/test_dotnet_surface.py is the part of this that runs without a .NET SDK.
SWAP="$WORK/swapped" && rm -rf "$SWAP" && mkdir -p "$SWAP" && cp -r "$SRC" "$SWAP/ICSharpCode.SharpZipLib"
[ -d "$SRC/../../assets" ] && cp -r "$SRC/../../assets" "$SWAP/../assets" 2>/dev/null || true
P="$SWAP/ICSharpCode.SharpZipLib"
rm "$P/Zip/Compression/Deflater.cs" "$P/Zip/Compression/Inflater.cs" "$P/Zip/Compression/Streams/InflaterInputStream.cs" "$P/Zip/Compression/Streams/DeflaterOutputStream.cs"
cp "$ROOT/sharpziplib_amd/dotnet/Deflater.Device.cs" "$P/Zip/Compression/"
cp "$ROOT/sharpziplib_amd/dotnet/InflaterInputStream.Device.cs" "$ROOT/sharpziplib_amd/dotnet/DeflaterOutputStream.Device.cs" "$P/Zip/Compression/Streams/"
dotnet build -c Release "$P/ICSharpCode.SharpZipLib.csproj" -p:AllowUnsafeBlocks=true -p:SignAssembly=false -p:TreatWarningsAsErrors=false \
  && echo "shim: the reference's assembly builds with the four files swapped"

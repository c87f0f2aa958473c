// RefGolden — the reference's own Deflater on the golden inputs (tools/dotnet_golden/README.md).
//   dotnet run -c Release -- <inputs dir> <out.json>
// <inputs dir>/cases.tsv lists one case per line:  name <TAB> file <TAB> level <TAB> mode
//   mode "raw"    : new Deflater(level, true); SetInput(all); Finish(); Deflate until IsFinished   (C/Deflater.cs:178,331,262,427)
//   mode "zlib"   : the same with new Deflater(level, false)
//   mode "stream" : DeflaterOutputStream(memory, new Deflater(level, true)), Write in 4096-byte pieces, Flush() in the middle, Finish()
//                   (T/Base/InflaterDeflaterTests.cs:64-69 is that call pattern; CS/DeflaterOutputStream.cs:388,100)
// Output: {"cases": {name: {"out_len": n, "out_sha256": hex, "seconds": s}}} — what tests/test_golden.py compares the oracle with.
using System;
using System.Collections.Generic;
using System.Diagnostics;
using System.Globalization;
using System.IO;
using System.Security.Cryptography;
using System.Text;
using ICSharpCode.SharpZipLib.Zip.Compression;
using ICSharpCode.SharpZipLib.Zip.Compression.Streams;

static class RefGolden
{
	sealed class HashSink : Stream
	{
		readonly IncrementalHash h = IncrementalHash.CreateHash(HashAlgorithmName.SHA256);
		public long Count;
		public override void Write(byte[] b, int o, int c) { h.AppendData(b, o, c); Count += c; }
		public string Hex() => Convert.ToHexString(h.GetHashAndReset()).ToLowerInvariant();
		public override bool CanRead => false; public override bool CanSeek => false; public override bool CanWrite => true;
		public override long Length => Count; public override long Position { get => Count; set => throw new NotSupportedException(); }
		public override void Flush() { } public override int Read(byte[] b, int o, int c) => throw new NotSupportedException();
		public override long Seek(long o, SeekOrigin s) => throw new NotSupportedException(); public override void SetLength(long v) => throw new NotSupportedException();
	}

	static void OneShot(byte[] data, int level, bool raw, HashSink sink)
	{
		var d = new Deflater(level, raw);
		var buf = new byte[1 << 20];
		// SetInput takes int counts: streams of 2 GiB and more go in as pieces (SURVEY App. A.6: chunk-independent at levels 5-9)
		long off = 0;
		do
		{
			int n = (int)Math.Min(data.LongLength - off, 1 << 30);
			d.SetInput(data, (int)off, n);
			off += n;
			if (off == data.LongLength) d.Finish();
			while (true)
			{
				int k = d.Deflate(buf, 0, buf.Length);
				if (k <= 0) break;
				sink.Write(buf, 0, k);
			}
		} while (off < data.LongLength);
		if (!d.IsFinished) throw new InvalidOperationException("Deflater did not finish");
	}

	static void Streamed(byte[] data, int level, HashSink sink)
	{
		var s = new DeflaterOutputStream(sink, new Deflater(level, true), 512) { IsStreamOwner = false };
		int half = data.Length / 2;
		for (int o = 0; o < data.Length; o += 4096)
		{
			if (o <= half && half < o + 4096) s.Flush();
			s.Write(data, o, Math.Min(4096, data.Length - o));
		}
		s.Finish();
	}

	static int Main(string[] args)
	{
		if (args.Length != 2) { Console.Error.WriteLine("usage: RefGolden <inputs dir> <out.json>"); return 2; }
		var sb = new StringBuilder("{\n \"_comment\": \"outputs of the reference's own Deflater (tools/dotnet_golden); compared with oracle/ by tests/test_golden.py\",\n \"cases\": {\n");
		bool first = true;
		foreach (string line in File.ReadAllLines(Path.Combine(args[0], "cases.tsv")))
		{
			if (line.Length == 0 || line[0] == '#') continue;
			string[] f = line.Split('\t');
			byte[] data = File.ReadAllBytes(Path.Combine(args[0], f[1]));
			int level = int.Parse(f[2], CultureInfo.InvariantCulture);
			var sink = new HashSink();
			var sw = Stopwatch.StartNew();
			switch (f[3])
			{
				case "raw": OneShot(data, level, true, sink); break;
				case "zlib": OneShot(data, level, false, sink); break;
				case "stream": Streamed(data, level, sink); break;
				default: throw new ArgumentException("mode " + f[3]);
			}
			sw.Stop();
			sb.Append(first ? "" : ",\n").AppendFormat(CultureInfo.InvariantCulture, "  \"{0}\": {{\"out_len\": {1}, \"out_sha256\": \"{2}\", \"seconds\": {3:F3}}}", f[0], sink.Count, sink.Hex(), sw.Elapsed.TotalSeconds);
			first = false;
			Console.Error.WriteLine("{0}: {1} -> {2} bytes in {3:F2} s ({4:F1} MiB/s)", f[0], data.LongLength, sink.Count, sw.Elapsed.TotalSeconds, data.LongLength / 1048576.0 / Math.Max(1e-9, sw.Elapsed.TotalSeconds));
		}
		sb.Append("\n }\n}\n");
		File.WriteAllText(args[1], sb.ToString());
		return 0;
	}
}

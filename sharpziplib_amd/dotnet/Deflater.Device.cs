// Deflater.Device.cs — drop-in replacement of ICSharpCode.SharpZipLib.Zip.Compression.Deflater
// (reference: src/ICSharpCode.SharpZipLib/Zip/Compression/Deflater.cs) that P/Invokes libszl_amd.so.
// Same namespace, same public members, same exceptions; DeflaterOutputStream / GZipOutputStream /
// ZipOutputStream keep using `new Deflater(level, nowrap)` and the protected `deflater_` field unchanged.
// NOTE: there is no .NET toolchain in the build image; this file is syntax-reviewed only.  The
// identical call sequences are exercised through the C ABI by tests/test_gpu_deflate.py.
using System;
using System.Runtime.InteropServices;

namespace ICSharpCode.SharpZipLib.Zip.Compression
{
	internal static class SzlNative
	{
		private const string Lib = "szl_amd"; // libszl_amd.so

		[DllImport(Lib)] internal static extern IntPtr szl_last_error();
		[DllImport(Lib)] internal static extern IntPtr szl_strerror(int status);

		[DllImport(Lib)] internal static extern IntPtr szl_deflater_create(int level, int noZlibHeaderOrFooter);
		[DllImport(Lib)] internal static extern void szl_deflater_destroy(IntPtr d);
		[DllImport(Lib)] internal static extern int szl_deflater_reset(IntPtr d);
		[DllImport(Lib)] internal static extern int szl_deflater_set_level(IntPtr d, int level);
		[DllImport(Lib)] internal static extern int szl_deflater_get_level(IntPtr d);
		[DllImport(Lib)] internal static extern int szl_deflater_set_strategy(IntPtr d, int strategy);
		[DllImport(Lib)] internal static extern unsafe int szl_deflater_set_dictionary(IntPtr d, byte* p, int n);
		[DllImport(Lib)] internal static extern unsafe int szl_deflater_set_input(IntPtr d, byte* p, int n);
		[DllImport(Lib)] internal static extern int szl_deflater_flush(IntPtr d);
		[DllImport(Lib)] internal static extern int szl_deflater_finish(IntPtr d);
		[DllImport(Lib)] internal static extern unsafe int szl_deflater_deflate(IntPtr d, byte* output, int len);
		[DllImport(Lib)] internal static extern unsafe int szl_deflater_deflate_view(IntPtr d, byte** p, long* n);
		[DllImport(Lib)] internal static extern int szl_deflater_needs_input(IntPtr d);
		[DllImport(Lib)] internal static extern int szl_deflater_is_finished(IntPtr d);
		[DllImport(Lib)] internal static extern long szl_deflater_total_in(IntPtr d);
		[DllImport(Lib)] internal static extern long szl_deflater_total_out(IntPtr d);
		[DllImport(Lib)] internal static extern uint szl_deflater_adler(IntPtr d);
		[DllImport(Lib)] internal static extern int szl_deflater_enable_crc32(IntPtr d, int on);
		[DllImport(Lib)] internal static extern int szl_deflater_caller_drains(IntPtr d, int on);
		[DllImport(Lib)] internal static extern uint szl_deflater_crc32(IntPtr d);

		[DllImport(Lib)] internal static extern IntPtr szl_inflater_create(int noHeader);
		[DllImport(Lib)] internal static extern void szl_inflater_destroy(IntPtr s);
		[DllImport(Lib)] internal static extern int szl_inflater_reset(IntPtr s);
		[DllImport(Lib)] internal static extern unsafe int szl_inflater_set_input(IntPtr s, byte* p, int n);
		[DllImport(Lib)] internal static extern unsafe int szl_inflater_set_dictionary(IntPtr s, byte* p, int n);
		[DllImport(Lib)] internal static extern unsafe int szl_inflater_inflate(IntPtr s, byte* output, int count);
		[DllImport(Lib)] internal static extern int szl_inflater_needs_input(IntPtr s);
		[DllImport(Lib)] internal static extern int szl_inflater_needs_dictionary(IntPtr s);
		[DllImport(Lib)] internal static extern int szl_inflater_is_finished(IntPtr s);
		[DllImport(Lib)] internal static extern int szl_inflater_remaining_input(IntPtr s);
		[DllImport(Lib)] internal static extern long szl_inflater_total_in(IntPtr s);
		[DllImport(Lib)] internal static extern long szl_inflater_total_out(IntPtr s);
		[DllImport(Lib)] internal static extern uint szl_inflater_adler(IntPtr s);
		[DllImport(Lib)] internal static extern int szl_inflater_detach_input(IntPtr s);
		[DllImport(Lib)] internal static extern int szl_inflater_expect_more(IntPtr s, int more);
		[DllImport(Lib)] internal static extern int szl_inflater_enable_crc32(IntPtr s, int on);
		[DllImport(Lib)] internal static extern uint szl_inflater_crc32(IntPtr s);

		// batch entry points (include/szl.h): the feed for ZipOutputStream.PutNextPassthroughEntry (INTEGRATION.md §3) and for
		// multi-member gzip; the *_multi_* forms spread the streams — or the position ranges of ONE long stream — over several GPUs
		[StructLayout(LayoutKind.Sequential)]
		internal struct SzlStream
		{
			public ulong in_off, in_len, out_off, out_cap, out_len;
			public uint crc32, adler32;
			public int status;
			public uint reserved;
			public ulong in_consumed;
		}
		internal const uint F_NOWRAP = 1, F_CRC32 = 2, F_ADLER32 = 4, F_SYNC_FLUSH_BEFORE_FINISH = 8, F_GZIP = 16;
		[DllImport(Lib)] internal static extern ulong szl_deflate_bound(ulong inLen);
		[DllImport(Lib)] internal static extern IntPtr szl_engine_create();
		[DllImport(Lib)] internal static extern void szl_engine_destroy(IntPtr e);
		[DllImport(Lib)] internal static extern unsafe int szl_deflate_batch_host(IntPtr e, byte* input, byte* output, SzlStream* streams, UIntPtr n, int level, int strategy, uint flags);
		[DllImport(Lib)] internal static extern unsafe int szl_inflate_batch_host(IntPtr e, byte* input, byte* output, SzlStream* streams, UIntPtr n, uint flags);
		[DllImport(Lib)] internal static extern unsafe int szl_deflate_batch_multi_host(int* devices, int nDev, byte* input, byte* output, SzlStream* streams, UIntPtr n, int level, int strategy, uint flags);
		[DllImport(Lib)] internal static extern unsafe int szl_inflate_batch_multi_host(int* devices, int nDev, byte* input, byte* output, SzlStream* streams, UIntPtr n, uint flags);

		// szl_status -> the exception the reference throws at the same place
		internal static Exception Map(int status, string what)
		{
			string detail = Marshal.PtrToStringAnsi(szl_last_error());
			string msg = Marshal.PtrToStringAnsi(szl_strerror(status));
			switch (status)
			{
				case -1: return new ArgumentOutOfRangeException(what, detail);          // SZL_E_ARG
				case -2: return new InvalidOperationException(detail ?? msg);             // SZL_E_STATE
				case -5: return new NotSupportedException(detail ?? msg);                 // SZL_E_UNSUPPORTED
				case -24: return new StreamDecodingException(msg);                         // SZL_E_DYN_HEADER
				case -28:                                                                  // SZL_E_INDEX (UpdateHash past the window array)
				case -27: return new IndexOutOfRangeException(msg);                        // SZL_E_CODE_OVERSUBSCRIBED (what BuildTree throws there)
				default: return new SharpZipBaseException(msg + (detail != null ? ": " + detail : "")); // incl. -3 device errors
			}
		}
	}

	public class Deflater : IDisposable
	{
		public const int BEST_COMPRESSION = 9, BEST_SPEED = 1, DEFAULT_COMPRESSION = -1, NO_COMPRESSION = 0, DEFLATED = 8;

		/// <summary>The constants again as an enum (Deflater.cs:92-122); FastZip.CompressionLevel is of this type (Zip/FastZip.cs:342,484,996).</summary>
		public enum CompressionLevel
		{
			BEST_COMPRESSION = Deflater.BEST_COMPRESSION,
			BEST_SPEED = Deflater.BEST_SPEED,
			DEFAULT_COMPRESSION = Deflater.DEFAULT_COMPRESSION,
			NO_COMPRESSION = Deflater.NO_COMPRESSION,
			DEFLATED = Deflater.DEFLATED
		}

		private IntPtr h;

		public Deflater() : this(DEFAULT_COMPRESSION, false) { }
		public Deflater(int level) : this(level, false) { }
		public Deflater(int level, bool noZlibHeaderOrFooter)
		{
			if (level != DEFAULT_COMPRESSION && (level < NO_COMPRESSION || level > BEST_COMPRESSION))
				throw new ArgumentOutOfRangeException(nameof(level));               // Deflater.cs:184-187
			h = SzlNative.szl_deflater_create(level, noZlibHeaderOrFooter ? 1 : 0);
			if (h == IntPtr.Zero) throw SzlNative.Map(-3, nameof(level));
		}

		public void Reset() { Check(SzlNative.szl_deflater_reset(h), nameof(Reset)); }                 // :204
		public int Adler => unchecked((int)SzlNative.szl_deflater_adler(h));                              // :215
		/// <summary>CRC-32 of the input given so far, kept on the device beside the compression (include/szl.h): a device-aware
		/// GZipOutputStream / ZipOutputStream switches it on before the first SetInput and reads it where the reference reads crc.Value
		/// (S/GZip/GzipOutputStream.cs:210,330; S/Zip/ZipOutputStream.cs:700).</summary>
		/// <summary>The caller takes everything Deflate() offers before it changes a parameter (DeflaterOutputStream does, :242-272).  Only then is a
		/// SetLevel / SetStrategy with 16 KiB or more pending answered (include/szl.h szl_deflater_caller_drains); otherwise NotSupportedException.</summary>
		public void CallerDrains(bool on = true) { Check(SzlNative.szl_deflater_caller_drains(h, on ? 1 : 0), nameof(CallerDrains)); }
		internal void EnableCrc32(bool on = true) { Check(SzlNative.szl_deflater_enable_crc32(h, on ? 1 : 0), nameof(EnableCrc32)); }
		internal long Crc32 => SzlNative.szl_deflater_crc32(h);
		public long TotalIn => SzlNative.szl_deflater_total_in(h);                                       // :226
		public long TotalOut => SzlNative.szl_deflater_total_out(h);                                     // :237
		public void Flush() { SzlNative.szl_deflater_flush(h); }                                         // :252
		public void Finish() { SzlNative.szl_deflater_finish(h); }                                       // :262
		public bool IsFinished => SzlNative.szl_deflater_is_finished(h) != 0;                             // :271
		public bool IsNeedingInput => SzlNative.szl_deflater_needs_input(h) != 0;                         // :285
		public void SetInput(byte[] input) { SetInput(input, 0, input.Length); }
		public unsafe void SetInput(byte[] input, int offset, int count)                                 // :331
		{
			if (input == null) throw new ArgumentNullException(nameof(input));
			if (offset < 0) throw new ArgumentOutOfRangeException(nameof(offset));
			if (count < 0 || offset > input.Length - count) throw new ArgumentOutOfRangeException(nameof(count));
			fixed (byte* p = input) Check(SzlNative.szl_deflater_set_input(h, p + offset, count), nameof(SetInput));
		}
		public void SetLevel(int level) { Check(SzlNative.szl_deflater_set_level(h, level), nameof(level)); } // :349
		public int GetLevel() => SzlNative.szl_deflater_get_level(h);                                     // :371
		public void SetStrategy(DeflateStrategy strategy) { Check(SzlNative.szl_deflater_set_strategy(h, (int)strategy), nameof(strategy)); } // :385
		public int Deflate(byte[] output) => Deflate(output, 0, output.Length);
		public unsafe int Deflate(byte[] output, int offset, int length)                                 // :427
		{
			// Deflater.cs:427 indexes output[offset..offset+length) itself (IndexOutOfRangeException when it does not fit);
			// native code must never see a range outside the managed array
			if (output == null) throw new ArgumentNullException(nameof(output));
			if (offset < 0 || length < 0 || offset > output.Length - length) throw new IndexOutOfRangeException();
			fixed (byte* p = output) return Check(SzlNative.szl_deflater_deflate(h, p + offset, length), nameof(Deflate));
		}
		// Deflate() without its copies (include/szl.h szl_deflater_deflate_view): everything the next Deflate() calls would hand out, in place —
		// the object's pinned output queue — valid until the next call on this object; false where Deflate() returns 0.  What the device-aware
		// DeflaterOutputStream (DeflaterOutputStream.Device.cs) writes to its base stream from.
		internal unsafe bool DeflateView(out byte* p, out long n)
		{
			byte* q; long k;
			Check(SzlNative.szl_deflater_deflate_view(h, &q, &k), nameof(Deflate));
			p = q; n = k;
			return k > 0;
		}
		public void SetDictionary(byte[] dictionary) { SetDictionary(dictionary, 0, dictionary.Length); }
		public unsafe void SetDictionary(byte[] dictionary, int index, int count)                        // :559
		{
			if (dictionary == null) throw new ArgumentNullException(nameof(dictionary));
			if (index < 0 || count < 0 || index > dictionary.Length - count) throw new ArgumentOutOfRangeException(nameof(count)); // DeflaterEngine.cs:198-229 reads buffer[offset..offset+length)
			fixed (byte* p = dictionary) Check(SzlNative.szl_deflater_set_dictionary(h, p + index, count), nameof(SetDictionary));
		}

		private static int Check(int status, string what) { if (status < 0) throw SzlNative.Map(status, what); return status; }
		public void Dispose() { if (h != IntPtr.Zero) { SzlNative.szl_deflater_destroy(h); h = IntPtr.Zero; } GC.SuppressFinalize(this); }
		~Deflater() { if (h != IntPtr.Zero) SzlNative.szl_deflater_destroy(h); }
	}

	public class Inflater : IDisposable
	{
		private IntPtr h;
		internal bool noHeader;   // read by InflaterPool.Return (src/ICSharpCode.SharpZipLib/Core/InflaterPool.cs:56)

		public Inflater() : this(false) { }
		public Inflater(bool noHeader)
		{
			this.noHeader = noHeader;
			h = SzlNative.szl_inflater_create(noHeader ? 1 : 0);
			if (h == IntPtr.Zero) throw SzlNative.Map(-3, nameof(noHeader));
		}
		public void Reset() { SzlNative.szl_inflater_reset(h); input = null; }                           // Inflater.cs:188
		public void SetInput(byte[] buffer) { SetInput(buffer, 0, buffer.Length); }
		public unsafe void SetInput(byte[] buffer, int index, int count)                                 // :629
		{
			if (buffer == null) throw new ArgumentNullException(nameof(buffer));                          // StreamManipulator.SetInput CS/StreamManipulator.cs:244-262
			if (index < 0) throw new ArgumentOutOfRangeException(nameof(index), "Cannot be negative");
			if (count < 0) throw new ArgumentOutOfRangeException(nameof(count), "Cannot be negative");
			if (index > buffer.Length - count) throw new ArgumentOutOfRangeException(nameof(count));
			fixed (byte* p = buffer) Check(SzlNative.szl_inflater_set_input(h, p + index, count), nameof(SetInput));
			// A long piece out of a buffer the device-aware InflaterInputBuffer pinned and registered (InflaterInputStream.Device.cs) is
			// read in place, as the reference's StreamManipulator keeps `window_ = buffer` (CS/StreamManipulator.cs:244-262): keep the
			// array reachable until the object has let go of it (IsNeedingInput, Reset, DetachInput).
			input = buffer;
		}
		private byte[] input;
		/// <summary>The object stops referring to the caller's buffer (what it has not consumed moves into its own memory):
		/// InflaterInputStream.Dispose calls this before a pooled Inflater outlives the stream's buffer.</summary>
		internal void DetachInput() { SzlNative.szl_inflater_detach_input(h); input = null; }
		/// <summary>Hint of the stream shim (include/szl.h szl_inflater_expect_more): true — the buffer just given was filled to the brim, more
		/// input follows, a parallel piece may end on its last block boundary and ask for input at once; false — the base stream has ended.
		/// Returns true if a remainder was waiting for input that will not come (the next Inflate() decodes it).</summary>
		internal bool ExpectMoreInput(bool more) => SzlNative.szl_inflater_expect_more(h, more ? 1 : 0) == 1;
		/// <summary>CRC-32 of the bytes handed out, kept on the device beside the decode (include/szl.h): a device-aware
		/// GZipInputStream / ZipInputStream switches it on before the first SetInput and reads it where the reference reads crc.Value
		/// (S/GZip/GzipInputStream.cs:141,337; S/Zip/ZipInputStream.cs:673).</summary>
		internal void EnableCrc32(bool on = true) { Check(SzlNative.szl_inflater_enable_crc32(h, on ? 1 : 0), nameof(EnableCrc32)); }
		internal long Crc32 => SzlNative.szl_inflater_crc32(h);
		public void SetDictionary(byte[] buffer) { SetDictionary(buffer, 0, buffer.Length); }
		public unsafe void SetDictionary(byte[] buffer, int index, int count)                            // :563
		{
			if (buffer == null) throw new ArgumentNullException(nameof(buffer));                          // Inflater.cs:565-571
			if (index < 0) throw new ArgumentOutOfRangeException(nameof(index));
			if (count < 0) throw new ArgumentOutOfRangeException(nameof(count));
			if (index > buffer.Length - count) throw new ArgumentOutOfRangeException(nameof(count));
			fixed (byte* p = buffer) Check(SzlNative.szl_inflater_set_dictionary(h, p + index, count), nameof(SetDictionary));
		}
		public int Inflate(byte[] buffer) => Inflate(buffer, 0, buffer.Length);
		public unsafe int Inflate(byte[] buffer, int offset, int count)                                  // :715
		{
			if (buffer == null) throw new ArgumentNullException(nameof(buffer));
			if (count < 0) throw new ArgumentOutOfRangeException(nameof(count), "count cannot be negative");
			if (offset < 0) throw new ArgumentOutOfRangeException(nameof(offset), "offset cannot be negative");
			if (offset + count > buffer.Length) throw new ArgumentException("count exceeds buffer bounds");
			fixed (byte* p = buffer) return Check(SzlNative.szl_inflater_inflate(h, p + offset, count), nameof(Inflate));
		}
		public bool IsNeedingInput => SzlNative.szl_inflater_needs_input(h) != 0;                          // :783
		public bool IsNeedingDictionary => SzlNative.szl_inflater_needs_dictionary(h) != 0;                // :794
		public bool IsFinished => SzlNative.szl_inflater_is_finished(h) != 0;                              // :806
		public int Adler => unchecked((int)SzlNative.szl_inflater_adler(h));                               // :823
		public long TotalOut => SzlNative.szl_inflater_total_out(h);                                      // :848
		public long TotalIn => SzlNative.szl_inflater_total_in(h);                                        // :862
		public int RemainingInput => SzlNative.szl_inflater_remaining_input(h);                           // :878

		private static int Check(int status, string what) { if (status < 0) throw SzlNative.Map(status, what); return status; }
		public void Dispose() { if (h != IntPtr.Zero) { SzlNative.szl_inflater_destroy(h); h = IntPtr.Zero; } GC.SuppressFinalize(this); }
		~Inflater() { if (h != IntPtr.Zero) SzlNative.szl_inflater_destroy(h); }
	}

	/// <summary>
	/// Many independent entries through ONE device call — the managed caller of the batch entry points.  A user who changes
	/// nothing reaches the device through <see cref="Deflater"/> one entry at a time (about 1.2 ms per 64 KiB entry: a chain of
	/// dependent kernels per entry, DESIGN.md §5); the same entries handed over together cost 0.04 ms each.  This class is that
	/// hand-over: it compresses the entries exactly as <c>new Deflater(level, true)</c> would (bit-identical) with CRC-32 computed
	/// on the device, and feeds an unchanged <c>ZipOutputStream</c> through <c>PutNextPassthroughEntry</c>
	/// (S/Zip/ZipOutputStream.cs:313-346), which writes the headers, descriptors and the central directory itself.
	/// </summary>
	public sealed class SzlBatch : IDisposable
	{
		private IntPtr engine;
		private readonly int[] devices;

		/// <param name="devices">GPU ordinals to spread the entries over (contiguous groups of about equal input bytes, one
		/// host thread and engine per device); null or empty: the current device.</param>
		public SzlBatch(int[] devices = null)
		{
			this.devices = devices != null && devices.Length > 0 ? (int[])devices.Clone() : null;
			if (this.devices == null)
			{
				engine = SzlNative.szl_engine_create();
				if (engine == IntPtr.Zero) throw SzlNative.Map(-3, "szl_engine_create");
			}
		}

		/// <summary>Raw-deflate every entry (level as in Deflater); returns the compressed bytes and fills crc32[i].
		/// The entries of one device call are packed into ONE managed byte[] each way, so a call is limited to 2 GiB of input and
		/// 2 GiB of worst-case output (szl_deflate_bound); a longer list is cut into consecutive calls below that limit.  A single
		/// entry whose bound alone exceeds it is an ArgumentException (feed it through the streaming Deflater instead).</summary>
		public byte[][] Deflate(System.Collections.Generic.IReadOnlyList<ArraySegment<byte>> entries, int level, out uint[] crc32)
		{
			const ulong Limit = 0x7FFFFFC7UL - 8UL;                              // largest byte[] the runtime allocates, minus the slack below
			int total = entries.Count;
			var all = new byte[total][];
			crc32 = new uint[total];
			int first = 0;
			while (first < total)
			{
				ulong inSum = 0, outSum = 0;
				int last = first;
				while (last < total)
				{
					ulong len = (ulong)entries[last].Count;
					ulong cap = (SzlNative.szl_deflate_bound(len) + 3UL) & ~3UL;
					if (cap > Limit) throw new ArgumentException("entry " + last + " is too long for one batched call (" + len + " bytes)", nameof(entries));
					if (last > first && (inSum + len > Limit || outSum + cap > Limit)) break;
					inSum += len; outSum += cap; last++;
				}
				var part = new ArraySegment<byte>[last - first];
				for (int i = first; i < last; i++) part[i - first] = entries[i];
				byte[][] got = DeflateOneCall(part, level, out uint[] crcs);
				Array.Copy(got, 0, all, first, got.Length);
				Array.Copy(crcs, 0, crc32, first, crcs.Length);
				first = last;
			}
			return all;
		}

		private unsafe byte[][] DeflateOneCall(System.Collections.Generic.IReadOnlyList<ArraySegment<byte>> entries, int level, out uint[] crc32)
		{
			int n = entries.Count;
			var st = new SzlNative.SzlStream[n];
			ulong inTotal = 0, outTotal = 0;
			for (int i = 0; i < n; i++)
			{
				ulong len = (ulong)entries[i].Count;
				ulong cap = (SzlNative.szl_deflate_bound(len) + 3UL) & ~3UL;     // output regions are 4-byte aligned (include/szl.h)
				st[i].in_off = inTotal; st[i].in_len = len; st[i].out_off = outTotal; st[i].out_cap = cap;
				inTotal += len; outTotal += cap;
			}
			var input = new byte[inTotal + 8];
			for (int i = 0; i < n; i++)
				Buffer.BlockCopy(entries[i].Array, entries[i].Offset, input, (int)st[i].in_off, entries[i].Count);
			var output = new byte[outTotal + 8];
			int rc;
			fixed (byte* pin = input, pout = output)
			fixed (SzlNative.SzlStream* ps = st)
			{
				if (devices == null)
					rc = SzlNative.szl_deflate_batch_host(engine, pin, pout, ps, (UIntPtr)(uint)n, level, 0, SzlNative.F_NOWRAP | SzlNative.F_CRC32);
				else
					fixed (int* pd = devices)
						rc = SzlNative.szl_deflate_batch_multi_host(pd, devices.Length, pin, pout, ps, (UIntPtr)(uint)n, level, 0, SzlNative.F_NOWRAP | SzlNative.F_CRC32);
			}
			if (rc < 0) throw SzlNative.Map(rc, "szl_deflate_batch");
			var result = new byte[n][];
			crc32 = new uint[n];
			for (int i = 0; i < n; i++)
			{
				if (st[i].status < 0) throw SzlNative.Map(st[i].status, "entry " + i);
				result[i] = new byte[st[i].out_len];
				Buffer.BlockCopy(output, (int)st[i].out_off, result[i], 0, (int)st[i].out_len);
				crc32[i] = st[i].crc32;
			}
			return result;
		}

		/// <summary>Compress the entries in one device call and write them to <paramref name="zip"/> as passthrough entries.</summary>
		public void WriteEntries(ICSharpCode.SharpZipLib.Zip.ZipOutputStream zip, System.Collections.Generic.IReadOnlyList<string> names,
		                         System.Collections.Generic.IReadOnlyList<ArraySegment<byte>> entries, int level = Deflater.DEFAULT_COMPRESSION)
		{
			byte[][] comp = Deflate(entries, level, out uint[] crc);
			for (int i = 0; i < comp.Length; i++)
			{
				var e = new ICSharpCode.SharpZipLib.Zip.ZipEntry(names[i])
				{
					CompressionMethod = ICSharpCode.SharpZipLib.Zip.CompressionMethod.Deflated,
					Crc = crc[i], Size = entries[i].Count, CompressedSize = comp[i].Length
				};
				zip.PutNextPassthroughEntry(e);                                   // S/Zip/ZipOutputStream.cs:313
				zip.Write(comp[i], 0, comp[i].Length);                            // already-deflated bytes
				zip.CloseEntry();
			}
		}

		public void Dispose() { if (engine != IntPtr.Zero) { SzlNative.szl_engine_destroy(engine); engine = IntPtr.Zero; } GC.SuppressFinalize(this); }
		~SzlBatch() { if (engine != IntPtr.Zero) SzlNative.szl_engine_destroy(engine); }
	}
}

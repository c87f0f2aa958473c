// InflaterInputStream.Device.cs — device-aware InflaterInputBuffer / InflaterInputStream: replaces
// src/ICSharpCode.SharpZipLib/Zip/Compression/Streams/InflaterInputStream.cs (both classes live in that one file).
//
// Why a third file (INTEGRATION.md §1): the reference's buffer class reads `bufferSize` bytes per Fill() — 4096 by default
// (CS/InflaterInputStream.cs:22, :342-358; GZipInputStream passes 4096, S/GZip/GzipInputStream.cs:72; ZipInputStream the same,
// S/Zip/ZipInputStream.cs:92-114) — and hands them to Inflater.SetInput.  The device Inflater decodes a 4 KiB piece with ONE
// wavefront (17 MiB/s); what it decodes with the whole chip is a piece of megabytes (include/szl.h, DESIGN.md §4.5).  So the
// buffer behind RawData is at least ReadAheadBytes long (16 MiB) whatever the constructor was given — `bufferSize` keeps the
// meaning the reference gives it, a lower bound (:35-38) — it is pinned and registered with the device runtime, and a
// SetInput out of it costs no host copy (szl_host_register).  Everything else is the reference's member set with the
// reference's meaning: GZipInputStream.ReadFooter (:305-351: Available += RemainingInput, ReadClearTextBuffer) and
// ZipInputStream (:443-444, :709) still find their trailers in that same buffer.
//
// Same namespace, same public / protected members, same exception types and messages.  No .NET toolchain in the build image:
// syntax-reviewed only; sharpziplib_amd/streams.py is this file member for member, and tests/test_gpu_read_ahead.py,
// test_gpu_gzip.py, test_gpu_inflate.py drive it on the device.
using System;
using System.IO;
using System.Runtime.InteropServices;
using System.Security.Cryptography;
using ICSharpCode.SharpZipLib.Core;

namespace ICSharpCode.SharpZipLib.Zip.Compression.Streams
{
	internal static class SzlHost
	{
		private const string Lib = "szl_amd";
		[DllImport(Lib)] internal static extern unsafe int szl_host_register(byte* p, UIntPtr n);
		[DllImport(Lib)] internal static extern unsafe int szl_host_unregister(byte* p);
	}

	public class InflaterInputBuffer
	{
		/// <summary>Bytes read from the base stream per Fill() unless the constructor asks for more; 0 = the reference's sizes.</summary>
		public static int ReadAheadBytes { get; set; } = 16 << 20;
		/// <summary>What a base stream that can tell it holds eight read-aheads or more gets instead (a piece of 64 MiB costs the device
		/// little more than one of 16 MiB: 2.8 -> 5.2 GiB/s, profiles/r05/read_path.log).</summary>
		public static int ReadAheadLongBytes { get; set; } = 64 << 20;

		public InflaterInputBuffer(Stream stream) : this(stream, 4096) { }                       // :22

		public InflaterInputBuffer(Stream stream, int bufferSize)                               // :32-41
		{
			inputStream = stream;
			if (bufferSize < 1024) bufferSize = 1024;
			long size = Math.Max(bufferSize, ReadAheadBytes);
			if (size > bufferSize && stream.CanSeek)
			{
				// no more than the base stream still holds (+1: the Fill() that meets the end sees it)
				long left = stream.Length - stream.Position;
				if (left >= 8L * ReadAheadBytes) size = Math.Max(size, ReadAheadLongBytes);
				size = Math.Max(bufferSize, Math.Min(size, left + 1));
			}
			rawData = NewBuffer((int)size, out rawPin);
			clearText = rawData;
		}

		// A pinned array the device can read by DMA.  GC.AllocateUninitializedArray(pinned: true) keeps its address for life;
		// registration tells the HIP runtime about it.  Unregistered by Dispose() (below) — the array itself stays valid for as long
		// as anybody (an Inflater that still reads it) refers to it, so a late SetInput merely goes through the copying path.
		private static unsafe byte[] NewBuffer(int size, out bool registered)
		{
			byte[] a = GC.AllocateUninitializedArray<byte>(size, pinned: true);
			registered = false;
			if (size >= (64 << 10))
				fixed (byte* p = a) registered = SzlHost.szl_host_register(p, (UIntPtr)(uint)size) == 0;
			return a;
		}

		// Unregistered by Dispose() only (InflaterInputStream.Dispose calls it), after the Inflater this buffer last fed has let go of the
		// borrowed pointer.  There is NO finalizer: a finalizer runs on its own thread in no order relative to that Inflater's, and an
		// unregister there would pull the pages from under a native object that still reads them by DMA (round-5 ADVICE).  A stream that is
		// dropped without Dispose() leaves its two registrations in place: the arrays live on the pinned object heap, whose memory the
		// runtime does not return, so the cost is two entries in the HIP runtime's table.
		private Inflater lastFed;
		private bool disposed;
		public unsafe void Dispose()
		{
			if (disposed) return;
			disposed = true;
			lastFed?.DetachInput();
			lastFed = null;
			if (rawPin) fixed (byte* p = rawData) SzlHost.szl_host_unregister(p);
			if (clearPin) fixed (byte* p = internalClearText) SzlHost.szl_host_unregister(p);
			rawPin = clearPin = false;
		}

		public int RawLength => rawLength;                                                      // :47
		public byte[] RawData => rawData;                                                       // :59
		public int ClearTextLength => clearTextLength;                                          // :70
		public byte[] ClearText => clearText;                                                   // :81
		public int Available { get { return available; } set { available = value; } }           // :93

		public void SetInflaterInput(Inflater inflater)                                         // :103
		{
			if (available > 0)
			{
				inflater.SetInput(clearText, clearTextLength - available, available);
				lastFed = inflater;
				available = 0;
				inflater.ExpectMoreInput(rawLength == rawData.Length);   // a buffer filled to the brim promises more (include/szl.h)
			}
		}

		public void Fill()                                                                      // :115
		{
			rawLength = 0;
			int toRead = rawData.Length;
			while (toRead > 0 && inputStream.CanRead)
			{
				int count = inputStream.Read(rawData, rawLength, toRead);
				if (count <= 0) break;
				rawLength += count;
				toRead -= count;
			}
			clearTextLength = cryptoTransform != null ? cryptoTransform.TransformBlock(rawData, 0, rawLength, clearText, 0) : rawLength;
			available = clearTextLength;
		}

		public int ReadRawBuffer(byte[] buffer) => ReadRawBuffer(buffer, 0, buffer.Length);     // :148
		public int ReadRawBuffer(byte[] outBuffer, int offset, int length) => ReadFrom(false, outBuffer, offset, length);     // :160
		public int ReadClearTextBuffer(byte[] outBuffer, int offset, int length) => ReadFrom(true, outBuffer, offset, length); // :195

		private int ReadFrom(bool clear, byte[] outBuffer, int offset, int length)
		{
			if (length < 0) throw new ArgumentOutOfRangeException(nameof(length));
			int at = offset, left = length;
			while (left > 0)
			{
				if (available <= 0)
				{
					Fill();
					if (available <= 0) return 0;
				}
				int n = Math.Min(left, available);
				if (clear) Array.Copy(clearText, clearTextLength - available, outBuffer, at, n);
				else Array.Copy(rawData, rawLength - available, outBuffer, at, n);
				at += n; left -= n; available -= n;
			}
			return length;
		}

		public byte ReadLeByte()                                                                // :232
		{
			if (available <= 0)
			{
				Fill();
				if (available <= 0) throw new ZipException("EOF in header");
			}
			byte result = rawData[rawLength - available];
			available -= 1;
			return result;
		}
		public int ReadLeShort() => ReadLeByte() | (ReadLeByte() << 8);                         // :251
		public int ReadLeInt() => ReadLeShort() | (ReadLeShort() << 16);                        // :260
		public long ReadLeLong() => (uint)ReadLeInt() | ((long)ReadLeInt() << 32);              // :269

		public ICryptoTransform CryptoTransform                                                 // :276-305
		{
			set
			{
				cryptoTransform = value;
				if (cryptoTransform != null)
				{
					if (rawData == clearText)
					{
						if (internalClearText == null) internalClearText = NewBuffer(rawData.Length, out clearPin);
						clearText = internalClearText;
					}
					clearTextLength = rawLength;
					if (available > 0)
						cryptoTransform.TransformBlock(rawData, rawLength - available, available, clearText, rawLength - available);
				}
				else
				{
					clearText = rawData;
					clearTextLength = rawLength;
				}
			}
		}

		private int rawLength;
		private readonly byte[] rawData;
		private int clearTextLength;
		private byte[] clearText;
		private byte[] internalClearText;
		private int available;
		private ICryptoTransform cryptoTransform;
		private readonly Stream inputStream;
		private bool rawPin;
		private bool clearPin;
	}

	public class InflaterInputStream : Stream
	{
		public InflaterInputStream(Stream baseInputStream) : this(baseInputStream, InflaterPool.Instance.Rent(), 4096) { }   // :342
		public InflaterInputStream(Stream baseInputStream, Inflater inf) : this(baseInputStream, inf, 4096) { }               // :357
		public InflaterInputStream(Stream baseInputStream, Inflater inflater, int bufferSize)                                 // :376
		{
			if (baseInputStream == null) throw new ArgumentNullException(nameof(baseInputStream));
			if (inflater == null) throw new ArgumentNullException(nameof(inflater));
			if (bufferSize <= 0) throw new ArgumentOutOfRangeException(nameof(bufferSize));
			this.baseInputStream = baseInputStream;
			this.inf = inflater;
			inputBuffer = new InflaterInputBuffer(baseInputStream, bufferSize);
		}

		public bool IsStreamOwner { get; set; } = true;                                         // :405

		public long Skip(long count)                                                            // :420
		{
			if (count <= 0) throw new ArgumentOutOfRangeException(nameof(count));
			if (baseInputStream.CanSeek)
			{
				baseInputStream.Seek(count, SeekOrigin.Current);
				return count;
			}
			int length = (int)Math.Min(2048, count);
			byte[] tmp = new byte[length];
			long toSkip = count;
			int readCount = 1;
			while (toSkip > 0 && readCount > 0)
			{
				if (toSkip < length) length = (int)toSkip;
				readCount = baseInputStream.Read(tmp, 0, length);
				toSkip -= readCount;
			}
			return count - toSkip;
		}

		protected void StopDecrypting() { inputBuffer.CryptoTransform = null; }                 // :468
		public virtual int Available => inf.IsFinished ? 0 : 1;                                 // :472

		protected void Fill()                                                                   // :486
		{
			if (inputBuffer.Available <= 0)
			{
				inputBuffer.Fill();
				if (inputBuffer.Available <= 0)
				{
					// The base stream has ended: the promise of more input is taken back.  If the remainder of the last parallel piece was
					// waiting for it, Inflate() decodes it now and Read() goes on — a truncated stream delivers every byte it holds before
					// "Unexpected EOF", as the reference's does.
					if (inf.ExpectMoreInput(false)) return;
					throw new SharpZipBaseException("Unexpected EOF");
				}
			}
			inputBuffer.SetInflaterInput(inf);
		}

		public override bool CanRead => baseInputStream.CanRead;                                // :507
		public override bool CanSeek => false;
		public override bool CanWrite => false;
		public override long Length => throw new NotSupportedException("InflaterInputStream Length is not supported");
		public override long Position
		{
			get { return baseInputStream.Position; }
			set { throw new NotSupportedException("InflaterInputStream Position not supported"); }
		}
		public override void Flush() { baseInputStream.Flush(); }
		public override long Seek(long offset, SeekOrigin origin) { throw new NotSupportedException("Seek not supported"); }
		public override void SetLength(long value) { throw new NotSupportedException("InflaterInputStream SetLength not supported"); }
		public override void Write(byte[] buffer, int offset, int count) { throw new NotSupportedException("InflaterInputStream Write not supported"); }
		public override void WriteByte(byte value) { throw new NotSupportedException("InflaterInputStream WriteByte not supported"); }

		protected override void Dispose(bool disposing)                                         // :622
		{
			if (!isClosed)
			{
				isClosed = true;
				if (IsStreamOwner) baseInputStream.Dispose();
			}
			// (a pooled Inflater lives on: it must stop referring to this stream's buffer — Inflater.DetachInput, Deflater.Device.cs)
			inf?.DetachInput();
			inputBuffer?.Dispose();                                                            // (unregisters its arrays: InflaterInputBuffer.Dispose)
			if (inf is PooledInflater inflater) InflaterPool.Instance.Return(inflater);
			inf = null;
		}

		public override int Read(byte[] buffer, int offset, int count)                          // :658
		{
			if (inf.IsNeedingDictionary) throw new SharpZipBaseException("Need a dictionary");
			int remainingBytes = count;
			while (true)
			{
				int bytesRead = inf.Inflate(buffer, offset, remainingBytes);
				offset += bytesRead;
				remainingBytes -= bytesRead;
				if (remainingBytes == 0 || inf.IsFinished) break;
				if (inf.IsNeedingInput) Fill();
				else if (bytesRead == 0) throw new ZipException("Invalid input data");
			}
			return count - remainingBytes;
		}

		protected Inflater inf;
		protected InflaterInputBuffer inputBuffer;
		private readonly Stream baseInputStream;
		protected long csize;
		private bool isClosed;
	}
}

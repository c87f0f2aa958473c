// DeflaterOutputStream.Device.cs — device-aware DeflaterOutputStream: replaces
// src/ICSharpCode.SharpZipLib/Zip/Compression/Streams/DeflaterOutputStream.cs (INTEGRATION.md §1, file 4).
//
// Why: the reference moves compressed bytes from the Deflater to the base stream buffer_.Length at a time — 512 bytes by default
// (CS/DeflaterOutputStream.cs:26-29, :41-44), 4096 under GZipOutputStream (S/GZip/GzipOutputStream.cs:72) and ZipOutputStream — one
// Deflate() and one base-stream Write per piece (:100-118, :242-272).  The device Deflater compresses at Flush() / Finish() and
// then HOLDS the whole result in pinned host memory the device wrote by DMA: handing a gigabyte's 380 MB out through a 512-byte
// array is 740 000 P/Invokes, copies and Writes.  Here HandOut() asks the Deflater for all it has, in place
// (szl_deflater_deflate_view, include/szl.h), and writes that to the base stream in one call.  With a crypto transform the bytes go
// through buffer_ exactly as in the reference, block by block, because TransformBlock works in place on the stream's own array
// (:227-231, :256) — ZipOutputStream's encrypted entries see the same block structure as before.
//
// Same namespace, same public / protected members, same exception types and messages.  No .NET toolchain in the build image:
// syntax-reviewed only; sharpziplib_amd/streams.py is this file member for member and tests/test_gpu_write_path.py,
// test_gpu_gzip.py, test_crypto_hook.py drive it on the device.
using ICSharpCode.SharpZipLib.Encryption;
using System;
using System.IO;
using System.Security.Cryptography;
using System.Text;
using System.Threading;
using System.Threading.Tasks;

namespace ICSharpCode.SharpZipLib.Zip.Compression.Streams
{
	public class DeflaterOutputStream : Stream
	{
		public DeflaterOutputStream(Stream baseOutputStream) : this(baseOutputStream, new Deflater(), 512) { }                    // :26
		public DeflaterOutputStream(Stream baseOutputStream, Deflater deflater) : this(baseOutputStream, deflater, 512) { }      // :41

		public DeflaterOutputStream(Stream baseOutputStream, Deflater deflater, int bufferSize)                                  // :68-90
		{
			if (baseOutputStream == null) throw new ArgumentNullException(nameof(baseOutputStream));
			if (!baseOutputStream.CanWrite) throw new ArgumentException("Must support writing", nameof(baseOutputStream));
			if (bufferSize < 512) throw new ArgumentOutOfRangeException(nameof(bufferSize));
			baseOutputStream_ = baseOutputStream;
			buffer_ = new byte[bufferSize];                   // (only a crypto transform's blocks pass through it now)
			deflater_ = deflater ?? throw new ArgumentNullException(nameof(deflater));
			deflater_.CallerDrains();                      // Deflate() below runs until IsNeedingInput: a SetLevel in mid-stream is exact (include/szl.h)
		}

		// One turn of the reference's two loops: the next compressed bytes go to the base stream; false where Deflate() returned <= 0.
		private unsafe bool HandOut()
		{
			if (cryptoTransform_ == null)
			{
				if (!deflater_.DeflateView(out byte* p, out long n)) return false;
				// (the view is valid until the next call on deflater_; Stream.Write copies or consumes what it is given before it returns)
				while (n > 0)
				{
					int k = (int)Math.Min(n, 1 << 30);
#if NETSTANDARD2_1 || NETCOREAPP2_1_OR_GREATER
					baseOutputStream_.Write(new ReadOnlySpan<byte>(p, k));
#else
					using (var ums = new UnmanagedMemoryStream(p, k)) ums.CopyTo(baseOutputStream_, 1 << 20);
#endif
					p += k; n -= k;
				}
				return true;
			}
			int len = deflater_.Deflate(buffer_, 0, buffer_.Length);
			if (len <= 0) return false;
			EncryptBlock(buffer_, 0, len);                    // :111, :256
			baseOutputStream_.Write(buffer_, 0, len);
			return true;
		}

		// (the asynchronous forms copy a view into a rented array first: a pointer cannot cross an await)
		private async Task<bool> HandOutAsync(CancellationToken ct)
		{
			if (cryptoTransform_ == null)
			{
				byte[] piece = TakeView();
				if (piece == null) return false;
				await baseOutputStream_.WriteAsync(piece, 0, piece.Length, ct).ConfigureAwait(false);
				return true;
			}
			int len = deflater_.Deflate(buffer_, 0, buffer_.Length);
			if (len <= 0) return false;
			EncryptBlock(buffer_, 0, len);
			await baseOutputStream_.WriteAsync(buffer_, 0, len, ct).ConfigureAwait(false);
			return true;
		}

		private unsafe byte[] TakeView()
		{
			if (!deflater_.DeflateView(out byte* p, out long n)) return null;
			if (n > int.MaxValue - 64) throw new SharpZipBaseException("more than 2 GiB of compressed bytes in one flush: use the synchronous Flush / Finish");
			var piece = new byte[n];
			fixed (byte* q = piece) Buffer.MemoryCopy(p, q, n, n);
			return piece;
		}

		public virtual void Finish()                              // :100-132
		{
			deflater_.Finish();
			while (!deflater_.IsFinished)
			{
				if (!HandOut()) break;
			}
			if (!deflater_.IsFinished) throw new SharpZipBaseException("Can't deflate all input?");
			baseOutputStream_.Flush();
			ReleaseCrypto();
		}

		public virtual async Task FinishAsync(CancellationToken ct)    // :141-173
		{
			deflater_.Finish();
			while (!deflater_.IsFinished)
			{
				if (!await HandOutAsync(ct).ConfigureAwait(false)) break;
			}
			if (!deflater_.IsFinished) throw new SharpZipBaseException("Can't deflate all input?");
			await baseOutputStream_.FlushAsync(ct).ConfigureAwait(false);
			ReleaseCrypto();
		}

		private void ReleaseCrypto()                               // :122-130
		{
			if (cryptoTransform_ == null) return;
			GetAuthCodeIfAES();
			cryptoTransform_.Dispose();
			cryptoTransform_ = null;
		}

		public bool IsStreamOwner { get; set; } = true;           // :180
		public bool CanPatchEntries => baseOutputStream_.CanSeek;  // :185

		protected ICryptoTransform cryptoTransform_;               // :200
		protected byte[] AESAuthCode;                              // :205

		public Encoding ZipCryptoEncoding                          // :208
		{
			get => _stringCodec.ZipCryptoEncoding;
			set { _stringCodec = _stringCodec.WithZipCryptoEncoding(value); }
		}

		protected void EncryptBlock(byte[] buffer, int offset, int length)    // :227-231
		{
			if (cryptoTransform_ is null) return;
			cryptoTransform_.TransformBlock(buffer, 0, length, buffer, 0);
		}

		protected void Deflate()                                   // :242
		{
			while (!deflater_.IsNeedingInput)
			{
				if (!HandOut()) break;
			}
			if (!deflater_.IsNeedingInput) throw new SharpZipBaseException("DeflaterOutputStream can't deflate all input?");
		}

		private void DeflateFlushing()                             // DeflateSyncOrAsync(flushing: true) :245-272
		{
			while (HandOut()) { }
			if (!deflater_.IsNeedingInput) throw new SharpZipBaseException("DeflaterOutputStream can't deflate all input?");
		}

		private async Task DeflateAsync(bool flushing, CancellationToken ct)
		{
			while (flushing || !deflater_.IsNeedingInput)
			{
				if (!await HandOutAsync(ct).ConfigureAwait(false)) break;
			}
			if (!deflater_.IsNeedingInput) throw new SharpZipBaseException("DeflaterOutputStream can't deflate all input?");
		}

		public override bool CanRead => false;                     // :281
		public override bool CanSeek => false;                     // :293
		public override bool CanWrite => baseOutputStream_.CanWrite;   // :304
		public override long Length => baseOutputStream_.Length;   // :315
		public override long Position                              // :327
		{
			get => baseOutputStream_.Position;
			set => throw new NotSupportedException("Position property not supported");
		}
		public override long Seek(long offset, SeekOrigin origin) => throw new NotSupportedException("DeflaterOutputStream Seek not supported");
		public override void SetLength(long value) => throw new NotSupportedException("DeflaterOutputStream SetLength not supported");
		public override int ReadByte() => throw new NotSupportedException("DeflaterOutputStream ReadByte not supported");
		public override int Read(byte[] buffer, int offset, int count) => throw new NotSupportedException("DeflaterOutputStream Read not supported");

		public override void Flush()                               // :388
		{
			deflater_.Flush();
			DeflateFlushing();
			baseOutputStream_.Flush();
		}

		public override async Task FlushAsync(CancellationToken cancellationToken)    // :401
		{
			deflater_.Flush();
			await DeflateAsync(true, cancellationToken).ConfigureAwait(false);
			await baseOutputStream_.FlushAsync(cancellationToken).ConfigureAwait(false);
		}

		protected override void Dispose(bool disposing)            // :412-437
		{
			if (isClosed_) return;
			isClosed_ = true;
			try
			{
				Finish();
				ReleaseCrypto();
			}
			finally
			{
				if (IsStreamOwner) baseOutputStream_.Dispose();
			}
		}

#if NETSTANDARD2_1 || NETCOREAPP3_0_OR_GREATER
		public override async ValueTask DisposeAsync()             // :443-467
		{
			if (isClosed_) return;
			isClosed_ = true;
			try
			{
				await FinishAsync(CancellationToken.None).ConfigureAwait(false);
				ReleaseCrypto();
			}
			finally
			{
				if (IsStreamOwner) await baseOutputStream_.DisposeAsync().ConfigureAwait(false);
			}
		}
#endif

		protected void GetAuthCodeIfAES()                          // :473
		{
			if (cryptoTransform_ is ZipAESTransform aes) AESAuthCode = aes.GetAuthCode();
		}

		public override void WriteByte(byte value)                 // :487
		{
			Write(new[] { value }, 0, 1);
		}

		public override void Write(byte[] buffer, int offset, int count)    // :506
		{
			deflater_.SetInput(buffer, offset, count);
			Deflate();
		}

		public override async Task WriteAsync(byte[] buffer, int offset, int count, CancellationToken ct)    // :527
		{
			deflater_.SetInput(buffer, offset, count);
			await DeflateAsync(false, ct).ConfigureAwait(false);
		}

		private byte[] buffer_;                                    // :541
		protected Deflater deflater_;                              // :546
		protected Stream baseOutputStream_;                        // :551
		private bool isClosed_;                                    // :553
		protected StringCodec _stringCodec = ZipStrings.GetStringCodec();   // :556
	}
}

"""A second, independent restatement of the reference's encoder — TEST INFRASTRUCTURE ONLY.

Written from the C# sources alone (/root/reference/src/ICSharpCode.SharpZipLib/Zip/Compression/: Deflater.cs,
DeflaterEngine.cs, DeflaterHuffman.cs, PendingBuffer.cs, DeflaterPending.cs, DeflaterConstants.cs), statement by
statement, in plain Python — NOT from oracle/*.c — so that an error in how the C oracle reads the reference does not
silently become the definition of "bit-exact" (the reference ships no expected compressed bytes and
cannot run here).  tests/test_cleanroom.py diffs the two restatements on the golden cases and on a thousand structured
random inputs.  Slow (pure Python loops): tens of KB per second.

C# semantics that matter and are kept: `short` arrays hold 16-bit two's-complement values (read back with & 0xffff where
the reference does), integer division truncates toward zero, `uint bits` in PendingBuffer, PendingBuffer.Reset() does
not clear `bits` (PendingBuffer.cs:43).
"""
import zlib

# ---- DeflaterConstants.cs
STORED_BLOCK, STATIC_TREES, DYN_TREES, PRESET_DICT = 0, 1, 2, 0x20
DEFAULT_MEM_LEVEL = 8
MAX_MATCH, MIN_MATCH, MAX_WBITS = 258, 3, 15
WSIZE = 1 << MAX_WBITS
WMASK = WSIZE - 1
HASH_BITS = DEFAULT_MEM_LEVEL + 7
HASH_SIZE = 1 << HASH_BITS
HASH_MASK = HASH_SIZE - 1
HASH_SHIFT = (HASH_BITS + MIN_MATCH - 1) // MIN_MATCH
MIN_LOOKAHEAD = MAX_MATCH + MIN_MATCH + 1
MAX_DIST = WSIZE - MIN_LOOKAHEAD
PENDING_BUF_SIZE = 1 << (DEFAULT_MEM_LEVEL + 8)
MAX_BLOCK_SIZE = min(65535, PENDING_BUF_SIZE - 5)
DEFLATE_STORED, DEFLATE_FAST, DEFLATE_SLOW = 0, 1, 2
GOOD_LENGTH = [0, 4, 4, 4, 4, 8, 8, 8, 32, 32]
MAX_LAZY = [0, 4, 5, 6, 4, 16, 16, 32, 128, 258]
NICE_LENGTH = [0, 8, 16, 32, 16, 32, 128, 128, 258, 258]
MAX_CHAIN = [0, 4, 8, 32, 16, 32, 128, 256, 1024, 4096]
COMPR_FUNC = [0, 1, 1, 1, 1, 2, 2, 2, 2, 2]


def _short(v):          # (short)v
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def _cdiv(a, b):        # C# integer division (truncates toward zero)
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


class PendingBuffer:    # PendingBuffer.cs
    def __init__(self, bufferSize=4096):
        self.buffer = bytearray(bufferSize)
        self.start = self.end = 0
        self.bits = 0       # uint
        self.bitCount = 0

    def Reset(self):        # :43 — `bits` is NOT cleared
        self.start = self.end = self.bitCount = 0

    def WriteShort(self, value):
        self.buffer[self.end] = value & 0xFF; self.end += 1
        self.buffer[self.end] = (value >> 8) & 0xFF; self.end += 1

    def WriteBlock(self, block, offset, length):
        self.buffer[self.end:self.end + length] = block[offset:offset + length]
        self.end += length

    @property
    def BitCount(self):
        return self.bitCount

    def AlignToByte(self):
        if self.bitCount > 0:
            self.buffer[self.end] = self.bits & 0xFF; self.end += 1
            if self.bitCount > 8:
                self.buffer[self.end] = (self.bits >> 8) & 0xFF; self.end += 1
        self.bits = 0
        self.bitCount = 0

    def WriteBits(self, b, count):
        self.bits = (self.bits | ((b << self.bitCount) & 0xFFFFFFFF)) & 0xFFFFFFFF
        self.bitCount += count
        if self.bitCount >= 16:
            self.buffer[self.end] = self.bits & 0xFF; self.end += 1
            self.buffer[self.end] = (self.bits >> 8) & 0xFF; self.end += 1
            self.bits >>= 16
            self.bitCount -= 16

    def WriteShortMSB(self, s):
        self.buffer[self.end] = (s >> 8) & 0xFF; self.end += 1
        self.buffer[self.end] = s & 0xFF; self.end += 1

    @property
    def IsFlushed(self):
        return self.end == 0

    def Flush(self, output, offset, length):
        if self.bitCount >= 8:
            self.buffer[self.end] = self.bits & 0xFF; self.end += 1
            self.bits >>= 8
            self.bitCount -= 8
        if length > self.end - self.start:
            length = self.end - self.start
            output[offset:offset + length] = self.buffer[self.start:self.start + length]
            self.start = 0
            self.end = 0
        else:
            output[offset:offset + length] = self.buffer[self.start:self.start + length]
            self.start += length
        return length


class DeflaterPending(PendingBuffer):   # DeflaterPending.cs
    def __init__(self):
        super().__init__(PENDING_BUF_SIZE)


# ---- DeflaterHuffman.cs
BUFSIZE = 1 << (DEFAULT_MEM_LEVEL + 6)
LITERAL_NUM, DIST_NUM, BITLEN_NUM = 286, 30, 19
REP_3_6, REP_3_10, REP_11_138 = 16, 17, 18
EOF_SYMBOL = 256
BL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
bit4Reverse = [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]


def BitReverse(toReverse):      # :924 (returns short)
    return _short(bit4Reverse[toReverse & 0xF] << 12 | bit4Reverse[(toReverse >> 4) & 0xF] << 8 |
                  bit4Reverse[(toReverse >> 8) & 0xF] << 4 | bit4Reverse[toReverse >> 12])


def Lcode(length):              # :932
    if length == 255:
        return 285
    code = 257
    while length >= 8:
        code += 4
        length >>= 1
    return code + length


def Dcode(distance):            # :948
    code = 0
    while distance >= 4:
        code += 2
        distance >>= 1
    return code + distance


class Tree:
    def __init__(self, dh, elems, minCodes, maxLength):
        self.dh = dh
        self.minNumCodes = minCodes
        self.maxLength = maxLength
        self.freqs = [0] * elems
        self.bl_counts = [0] * maxLength
        self.codes = None
        self.length = None
        self.numCodes = 0

    def Reset(self):
        for i in range(len(self.freqs)):
            self.freqs[i] = 0
        self.codes = None
        self.length = None

    def WriteSymbol(self, code):
        self.dh.pending.WriteBits(self.codes[code] & 0xFFFF, self.length[code])

    def SetStaticCodes(self, staticCodes, staticLengths):
        self.codes = staticCodes
        self.length = staticLengths

    def BuildCodes(self):       # :151
        nextCode = [0] * self.maxLength
        code = 0
        self.codes = [0] * len(self.freqs)
        for bits in range(self.maxLength):
            nextCode[bits] = code
            code += self.bl_counts[bits] << (15 - bits)
        for i in range(self.numCodes):
            bits = self.length[i]
            if bits > 0:
                self.codes[i] = BitReverse(nextCode[bits - 1])
                nextCode[bits - 1] += 1 << (16 - bits)

    def BuildTree(self):        # :196
        freqs = self.freqs
        numSymbols = len(freqs)
        heap = [0] * numSymbols
        heapLen = 0
        maxCode = 0
        for n in range(numSymbols):
            freq = freqs[n]
            if freq != 0:
                pos = heapLen
                heapLen += 1
                while pos > 0:
                    ppos = (pos - 1) // 2
                    if not (freqs[heap[ppos]] > freq):
                        break
                    heap[pos] = heap[ppos]
                    pos = ppos
                heap[pos] = n
                maxCode = n
        while heapLen < 2:
            if maxCode < 2:
                maxCode += 1
                node = maxCode
            else:
                node = 0
            heap[heapLen] = node
            heapLen += 1
        self.numCodes = max(maxCode + 1, self.minNumCodes)
        numLeafs = heapLen
        childs = [0] * (4 * heapLen - 2)
        values = [0] * (2 * heapLen - 1)
        numNodes = numLeafs
        for i in range(heapLen):
            node = heap[i]
            childs[2 * i] = node
            childs[2 * i + 1] = -1
            values[i] = freqs[node] << 8
            heap[i] = i
        while True:
            first = heap[0]
            heapLen -= 1
            last = heap[heapLen]
            ppos = 0
            path = 1
            while path < heapLen:
                if path + 1 < heapLen and values[heap[path]] > values[heap[path + 1]]:
                    path += 1
                heap[ppos] = heap[path]
                ppos = path
                path = path * 2 + 1
            lastVal = values[last]
            while True:             # while ((path = ppos) > 0 && values[heap[ppos = (path - 1) / 2]] > lastVal)
                path = ppos
                if not (path > 0):
                    break
                ppos = (path - 1) // 2
                if not (values[heap[ppos]] > lastVal):
                    break
                heap[path] = heap[ppos]
            heap[path] = last
            second = heap[0]
            last = numNodes
            numNodes += 1
            childs[2 * last] = first
            childs[2 * last + 1] = second
            mindepth = min(values[first] & 0xFF, values[second] & 0xFF)
            values[last] = lastVal = values[first] + values[second] - mindepth + 1
            ppos = 0
            path = 1
            while path < heapLen:
                if path + 1 < heapLen and values[heap[path]] > values[heap[path + 1]]:
                    path += 1
                heap[ppos] = heap[path]
                ppos = path
                path = ppos * 2 + 1
            while True:
                path = ppos
                if not (path > 0):
                    break
                ppos = (path - 1) // 2
                if not (values[heap[ppos]] > lastVal):
                    break
                heap[path] = heap[ppos]
            heap[path] = last
            if not (heapLen > 1):
                break
        if heap[0] != len(childs) // 2 - 1:
            raise RuntimeError("Heap invariant violated")
        self.BuildLength(childs)

    def GetEncodedLength(self):
        return sum(self.freqs[i] * self.length[i] for i in range(len(self.freqs)))

    def _scan(self, on_len, on_rep):
        """Shared control flow of CalcBLFreq (:349) and WriteTree (:411): they are the same loop with different actions."""
        curlen = -1
        i = 0
        length = self.length
        while i < self.numCodes:
            count = 1
            nextlen = length[i]
            if nextlen == 0:
                max_count, min_count = 138, 3
            else:
                max_count, min_count = 6, 3
                if curlen != nextlen:
                    on_len(nextlen, 1)
                    count = 0
            curlen = nextlen
            i += 1
            while i < self.numCodes and curlen == length[i]:
                i += 1
                count += 1
                if count >= max_count:
                    break
            if count < min_count:
                on_len(curlen, count)
            elif curlen != 0:
                on_rep(REP_3_6, count - 3, 2)
            elif count <= 10:
                on_rep(REP_3_10, count - 3, 3)
            else:
                on_rep(REP_11_138, count - 11, 7)

    def CalcBLFreq(self, blTree):
        def on_len(l, c):
            blTree.freqs[l] += c

        def on_rep(sym, extra, nbits):
            blTree.freqs[sym] += 1
        self._scan(on_len, on_rep)

    def WriteTree(self, blTree):
        def on_len(l, c):
            for _ in range(c):
                blTree.WriteSymbol(l)

        def on_rep(sym, extra, nbits):
            blTree.WriteSymbol(sym)
            self.dh.pending.WriteBits(extra, nbits)
        self._scan(on_len, on_rep)

    def BuildLength(self, childs):      # :475
        self.length = [0] * len(self.freqs)
        numNodes = len(childs) // 2
        numLeafs = (numNodes + 1) // 2
        overflow = 0
        maxLength = self.maxLength
        bl_counts = self.bl_counts
        for i in range(maxLength):
            bl_counts[i] = 0
        lengths = [0] * numNodes
        lengths[numNodes - 1] = 0
        for i in range(numNodes - 1, -1, -1):
            if childs[2 * i + 1] != -1:
                bitLength = lengths[i] + 1
                if bitLength > maxLength:
                    bitLength = maxLength
                    overflow += 1
                lengths[childs[2 * i]] = lengths[childs[2 * i + 1]] = bitLength
            else:
                bitLength = lengths[i]
                bl_counts[bitLength - 1] += 1
                self.length[childs[2 * i]] = lengths[i] & 0xFF
        if overflow == 0:
            return
        incrBitLen = maxLength - 1
        while True:
            while True:
                incrBitLen -= 1
                if bl_counts[incrBitLen] != 0:
                    break
            while True:
                bl_counts[incrBitLen] -= 1
                incrBitLen += 1
                bl_counts[incrBitLen] += 1
                overflow -= 1 << (maxLength - 1 - incrBitLen)
                if not (overflow > 0 and incrBitLen < maxLength - 1):
                    break
            if not (overflow > 0):
                break
        bl_counts[maxLength - 1] += overflow
        bl_counts[maxLength - 2] -= overflow
        nodePtr = 2 * numLeafs
        for bits in range(maxLength, 0, -1):
            n = bl_counts[bits - 1]
            while n > 0:
                childPtr = 2 * childs[nodePtr]
                nodePtr += 1
                if childs[childPtr + 1] == -1:
                    self.length[childs[childPtr]] = bits
                    n -= 1


def _static_tables():           # static DeflaterHuffman() :602
    lc, ll = [0] * LITERAL_NUM, [0] * LITERAL_NUM
    i = 0
    while i < 144:
        lc[i] = BitReverse((0x030 + i) << 8); ll[i] = 8; i += 1
    while i < 256:
        lc[i] = BitReverse((0x190 - 144 + i) << 7); ll[i] = 9; i += 1
    while i < 280:
        lc[i] = BitReverse((0x000 - 256 + i) << 9); ll[i] = 7; i += 1
    while i < LITERAL_NUM:
        lc[i] = BitReverse((0x0c0 - 280 + i) << 8); ll[i] = 8; i += 1
    dc, dl = [0] * DIST_NUM, [0] * DIST_NUM
    for i in range(DIST_NUM):
        dc[i] = BitReverse(i << 11); dl[i] = 5
    return lc, ll, dc, dl


staticLCodes, staticLLength, staticDCodes, staticDLength = _static_tables()


class DeflaterHuffman:
    def __init__(self, pending):
        self.pending = pending
        self.literalTree = Tree(self, LITERAL_NUM, 257, 15)
        self.distTree = Tree(self, DIST_NUM, 1, 15)
        self.blTree = Tree(self, BITLEN_NUM, 4, 7)
        self.d_buf = [0] * BUFSIZE
        self.l_buf = [0] * BUFSIZE
        self.last_lit = 0
        self.extra_bits = 0

    def Reset(self):
        self.last_lit = 0
        self.extra_bits = 0
        self.literalTree.Reset()
        self.distTree.Reset()
        self.blTree.Reset()

    def SendAllTrees(self, blTreeCodes):    # :676
        self.blTree.BuildCodes()
        self.literalTree.BuildCodes()
        self.distTree.BuildCodes()
        self.pending.WriteBits(self.literalTree.numCodes - 257, 5)
        self.pending.WriteBits(self.distTree.numCodes - 1, 5)
        self.pending.WriteBits(blTreeCodes - 4, 4)
        for rank in range(blTreeCodes):
            self.pending.WriteBits(self.blTree.length[BL_ORDER[rank]], 3)
        self.literalTree.WriteTree(self.blTree)
        self.distTree.WriteTree(self.blTree)

    def CompressBlock(self):    # :701
        for i in range(self.last_lit):
            litlen = self.l_buf[i] & 0xFF
            dist = self.d_buf[i]
            if dist != 0:
                dist -= 1
                lc = Lcode(litlen)
                self.literalTree.WriteSymbol(lc)
                bits = _cdiv(lc - 261, 4)
                if bits > 0 and bits <= 5:
                    self.pending.WriteBits(litlen & ((1 << bits) - 1), bits)
                dc = Dcode(dist)
                self.distTree.WriteSymbol(dc)
                bits = _cdiv(dc, 2) - 1
                if bits > 0:
                    self.pending.WriteBits(dist & ((1 << bits) - 1), bits)
            else:
                self.literalTree.WriteSymbol(litlen)
        self.literalTree.WriteSymbol(EOF_SYMBOL)

    def FlushStoredBlock(self, stored, storedOffset, storedLength, lastBlock):   # :766
        self.pending.WriteBits((STORED_BLOCK << 1) + (1 if lastBlock else 0), 3)
        self.pending.AlignToByte()
        self.pending.WriteShort(storedLength)
        self.pending.WriteShort(~storedLength)
        self.pending.WriteBlock(stored, storedOffset, storedLength)
        self.Reset()

    def FlushBlock(self, stored, storedOffset, storedLength, lastBlock):         # :788
        self.literalTree.freqs[EOF_SYMBOL] += 1
        self.literalTree.BuildTree()
        self.distTree.BuildTree()
        self.literalTree.CalcBLFreq(self.blTree)
        self.distTree.CalcBLFreq(self.blTree)
        self.blTree.BuildTree()
        blTreeCodes = 4
        for i in range(18, blTreeCodes, -1):
            if self.blTree.length[BL_ORDER[i]] > 0:
                blTreeCodes = i + 1     # NB the C# loop condition re-reads blTreeCodes: `for (i = 18; i > blTreeCodes; i--)`
                break                   # after the assignment i > i+1 is false, so the loop ends here
        opt_len = 14 + blTreeCodes * 3 + self.blTree.GetEncodedLength() + self.literalTree.GetEncodedLength() + \
            self.distTree.GetEncodedLength() + self.extra_bits
        static_len = self.extra_bits
        for i in range(LITERAL_NUM):
            static_len += self.literalTree.freqs[i] * staticLLength[i]
        for i in range(DIST_NUM):
            static_len += self.distTree.freqs[i] * staticDLength[i]
        if opt_len >= static_len:
            opt_len = static_len
        if storedOffset >= 0 and storedLength + 4 < opt_len >> 3:
            self.FlushStoredBlock(stored, storedOffset, storedLength, lastBlock)
        elif opt_len == static_len:
            self.pending.WriteBits((STATIC_TREES << 1) + (1 if lastBlock else 0), 3)
            self.literalTree.SetStaticCodes(staticLCodes, staticLLength)
            self.distTree.SetStaticCodes(staticDCodes, staticDLength)
            self.CompressBlock()
            self.Reset()
        else:
            self.pending.WriteBits((DYN_TREES << 1) + (1 if lastBlock else 0), 3)
            self.SendAllTrees(blTreeCodes)
            self.CompressBlock()
            self.Reset()

    def IsFull(self):
        return self.last_lit >= BUFSIZE

    def TallyLit(self, literal):    # :873
        self.d_buf[self.last_lit] = 0
        self.l_buf[self.last_lit] = literal & 0xFF
        self.last_lit += 1
        self.literalTree.freqs[literal] += 1
        return self.IsFull()

    def TallyDist(self, distance, length):  # :894
        self.d_buf[self.last_lit] = _short(distance)
        self.l_buf[self.last_lit] = (length - 3) & 0xFF
        self.last_lit += 1
        lc = Lcode(length - 3)
        self.literalTree.freqs[lc] += 1
        if lc >= 265 and lc < 285:
            self.extra_bits += _cdiv(lc - 261, 4)
        dc = Dcode(distance - 1)
        self.distTree.freqs[dc] += 1
        if dc >= 4:
            self.extra_bits += _cdiv(dc, 2) - 1
        return self.IsFull()


# ---- DeflaterEngine.cs
TooFar = 4096
Default, Filtered, HuffmanOnly = 0, 1, 2


class DeflaterEngine:
    def __init__(self, pending, noAdlerCalculation=False):
        self.pending = pending
        self.huffman = DeflaterHuffman(pending)
        self.adler = None if noAdlerCalculation else 1      # Adler32.Value (1 after Reset)
        self.window = bytearray(2 * WSIZE)
        self.head = [0] * HASH_SIZE     # short[]
        self.prev = [0] * WSIZE         # short[]
        self.blockStart = self.strstart = 1
        self.ins_h = 0
        self.matchStart = 0
        self.matchLen = 0
        self.prevAvailable = False
        self.lookahead = 0
        self.strategy = Default
        self.max_chain = self.max_lazy = self.niceLength = self.goodLength = 0
        self.compressionFunction = 0
        self.inputBuf = None
        self.totalIn = 0
        self.inputOff = 0
        self.inputEnd = 0

    def Deflate(self, flush, finish):       # :104
        while True:
            self.FillWindow()
            canFlush = flush and (self.inputOff == self.inputEnd)
            if self.compressionFunction == DEFLATE_STORED:
                progress = self.DeflateStored(canFlush, finish)
            elif self.compressionFunction == DEFLATE_FAST:
                progress = self.DeflateFast(canFlush, finish)
            elif self.compressionFunction == DEFLATE_SLOW:
                progress = self.DeflateSlow(canFlush, finish)
            else:
                raise RuntimeError("unknown compressionFunction")
            if not (self.pending.IsFlushed and progress):
                break
        return progress

    def SetInput(self, buffer, offset, count):  # :146
        if buffer is None:
            raise ValueError("buffer")
        if offset < 0:
            raise ValueError("offset")
        if count < 0:
            raise ValueError("count")
        if self.inputOff < self.inputEnd:
            raise RuntimeError("Old input was not completely processed")
        end = offset + count
        if offset > end or end > len(buffer):
            raise ValueError("count")
        self.inputBuf = buffer
        self.inputOff = offset
        self.inputEnd = end

    def NeedsInput(self):
        return self.inputEnd == self.inputOff

    def SetDictionary(self, buffer, offset, length):    # :198
        if self.adler is not None:
            self.adler = zlib.adler32(bytes(buffer[offset:offset + length]), self.adler)
        if length < MIN_MATCH:
            return
        if length > MAX_DIST:
            offset += length - MAX_DIST
            length = MAX_DIST
        self.window[self.strstart:self.strstart + length] = buffer[offset:offset + length]
        self.UpdateHash()
        length -= 1
        while True:
            length -= 1
            if not (length > 0):
                break
            self.InsertString()
            self.strstart += 1
        self.strstart += 2
        self.blockStart = self.strstart

    def Reset(self):    # :234
        self.huffman.Reset()
        if self.adler is not None:
            self.adler = 1
        self.blockStart = self.strstart = 1
        self.lookahead = 0
        self.totalIn = 0
        self.prevAvailable = False
        self.matchLen = MIN_MATCH - 1
        for i in range(HASH_SIZE):
            self.head[i] = 0
        for i in range(WSIZE):
            self.prev[i] = 0

    def ResetAdler(self):
        if self.adler is not None:
            self.adler = 1

    @property
    def Adler(self):
        return self.adler if self.adler is not None else 0

    @property
    def TotalIn(self):
        return self.totalIn

    def SetLevel(self, level):  # :304
        if level < 0 or level > 9:
            raise ValueError("level")
        self.goodLength = GOOD_LENGTH[level]
        self.max_lazy = MAX_LAZY[level]
        self.niceLength = NICE_LENGTH[level]
        self.max_chain = MAX_CHAIN[level]
        if COMPR_FUNC[level] != self.compressionFunction:
            cf = self.compressionFunction
            if cf == DEFLATE_STORED:
                if self.strstart > self.blockStart:
                    self.huffman.FlushStoredBlock(self.window, self.blockStart, self.strstart - self.blockStart, False)
                    self.blockStart = self.strstart
                self.UpdateHash()
            elif cf == DEFLATE_FAST:
                if self.strstart > self.blockStart:
                    self.huffman.FlushBlock(self.window, self.blockStart, self.strstart - self.blockStart, False)
                    self.blockStart = self.strstart
            elif cf == DEFLATE_SLOW:
                if self.prevAvailable:
                    self.huffman.TallyLit(self.window[self.strstart - 1] & 0xFF)
                if self.strstart > self.blockStart:
                    self.huffman.FlushBlock(self.window, self.blockStart, self.strstart - self.blockStart, False)
                    self.blockStart = self.strstart
                self.prevAvailable = False
                self.matchLen = MIN_MATCH - 1
            self.compressionFunction = COMPR_FUNC[level]

    def FillWindow(self):   # :366
        if self.strstart >= WSIZE + MAX_DIST:
            self.SlideWindow()
        if self.lookahead < MIN_LOOKAHEAD and self.inputOff < self.inputEnd:
            more = 2 * WSIZE - self.lookahead - self.strstart
            if more > self.inputEnd - self.inputOff:
                more = self.inputEnd - self.inputOff
            w0 = self.strstart + self.lookahead
            self.window[w0:w0 + more] = self.inputBuf[self.inputOff:self.inputOff + more]
            if self.adler is not None:
                self.adler = zlib.adler32(bytes(self.inputBuf[self.inputOff:self.inputOff + more]), self.adler)
            self.inputOff += more
            self.totalIn += more
            self.lookahead += more
        if self.lookahead >= MIN_MATCH:
            self.UpdateHash()

    def UpdateHash(self):   # :402
        self.ins_h = (self.window[self.strstart] << HASH_SHIFT) ^ self.window[self.strstart + 1]

    def InsertString(self):  # :417
        hash_ = ((self.ins_h << HASH_SHIFT) ^ self.window[self.strstart + (MIN_MATCH - 1)]) & HASH_MASK
        match = self.head[hash_]
        self.prev[self.strstart & WMASK] = match
        self.head[hash_] = _short(self.strstart)
        self.ins_h = hash_
        return match & 0xFFFF

    def SlideWindow(self):  # :441
        self.window[0:WSIZE] = self.window[WSIZE:2 * WSIZE]
        self.matchStart -= WSIZE
        self.strstart -= WSIZE
        self.blockStart -= WSIZE
        for i in range(HASH_SIZE):
            m = self.head[i] & 0xFFFF
            self.head[i] = _short(m - WSIZE if m >= WSIZE else 0)
        for i in range(WSIZE):
            m = self.prev[i] & 0xFFFF
            self.prev[i] = _short(m - WSIZE if m >= WSIZE else 0)

    def FindLongestMatch(self, curMatch):   # :474
        window = self.window
        prev = self.prev
        scan = self.strstart
        scanMax = scan + min(MAX_MATCH, self.lookahead) - 1
        limit = max(scan - MAX_DIST, 0)
        chainLength = self.max_chain
        niceLength = min(self.niceLength, self.lookahead)
        self.matchLen = max(self.matchLen, MIN_MATCH - 1)
        if scan + self.matchLen > scanMax:
            return False
        scan_end1 = window[scan + self.matchLen - 1]
        scan_end = window[scan + self.matchLen]
        if self.matchLen >= self.goodLength:
            chainLength >>= 2
        while True:
            match = curMatch
            scan = self.strstart
            skip = (window[match + self.matchLen] != scan_end or window[match + self.matchLen - 1] != scan_end1 or
                    window[match] != window[scan])
            if not skip:
                match += 1
                scan += 1
                skip = window[match] != window[scan]
            if not skip:
                # the switch on (scanMax - scan) % 8 (:515-565): compare that many further bytes, stopping at the first mismatch
                r = (scanMax - scan) % 8
                k = 0
                while k < r:
                    scan += 1
                    match += 1
                    if window[scan] != window[match]:
                        break
                    k += 1
                if window[scan] == window[match]:
                    while True:     # :573-590
                        if scan == scanMax:
                            scan += 1
                            match += 1
                            break
                        cont = True
                        for _ in range(8):
                            scan += 1
                            match += 1
                            if window[scan] != window[match]:
                                cont = False
                                break
                        if not cont:
                            break
                if scan - self.strstart > self.matchLen:
                    self.matchStart = curMatch
                    self.matchLen = scan - self.strstart
                    if self.matchLen >= niceLength:
                        break
                    scan_end1 = window[scan - 1]
                    scan_end = window[scan]
            curMatch = prev[curMatch & WMASK] & 0xFFFF
            if not (curMatch > limit):
                break
            chainLength -= 1
            if chainLength == 0:
                break
        return self.matchLen >= MIN_MATCH

    def DeflateStored(self, flush, finish):     # :614
        if not flush and self.lookahead == 0:
            return False
        self.strstart += self.lookahead
        self.lookahead = 0
        storedLength = self.strstart - self.blockStart
        if storedLength >= MAX_BLOCK_SIZE or (self.blockStart < WSIZE and storedLength >= MAX_DIST) or flush:
            lastBlock = finish
            if storedLength > MAX_BLOCK_SIZE:
                storedLength = MAX_BLOCK_SIZE
                lastBlock = False
            self.huffman.FlushStoredBlock(self.window, self.blockStart, storedLength, lastBlock)
            self.blockStart += storedLength
            return not (lastBlock or storedLength == 0)
        return True

    def DeflateFast(self, flush, finish):       # :651
        if self.lookahead < MIN_LOOKAHEAD and not flush:
            return False
        while self.lookahead >= MIN_LOOKAHEAD or flush:
            if self.lookahead == 0:
                self.huffman.FlushBlock(self.window, self.blockStart, self.strstart - self.blockStart, finish)
                self.blockStart = self.strstart
                return False
            if self.strstart > 2 * WSIZE - MIN_LOOKAHEAD:
                self.SlideWindow()
            found = False
            if self.lookahead >= MIN_MATCH:
                hashHead = self.InsertString()
                if hashHead != 0 and self.strategy != HuffmanOnly and self.strstart - hashHead <= MAX_DIST and \
                        self.FindLongestMatch(hashHead):
                    found = True
            if found:
                full = self.huffman.TallyDist(self.strstart - self.matchStart, self.matchLen)
                self.lookahead -= self.matchLen
                if self.matchLen <= self.max_lazy and self.lookahead >= MIN_MATCH:
                    while True:
                        self.matchLen -= 1
                        if not (self.matchLen > 0):
                            break
                        self.strstart += 1
                        self.InsertString()
                    self.strstart += 1
                else:
                    self.strstart += self.matchLen
                    if self.lookahead >= MIN_MATCH - 1:
                        self.UpdateHash()
                self.matchLen = MIN_MATCH - 1
                if not full:
                    continue
            else:
                self.huffman.TallyLit(self.window[self.strstart] & 0xFF)
                self.strstart += 1
                self.lookahead -= 1
            if self.huffman.IsFull():
                lastBlock = finish and (self.lookahead == 0)
                self.huffman.FlushBlock(self.window, self.blockStart, self.strstart - self.blockStart, lastBlock)
                self.blockStart = self.strstart
                return not lastBlock
        return True

    def DeflateSlow(self, flush, finish):       # :741
        if self.lookahead < MIN_LOOKAHEAD and not flush:
            return False
        while self.lookahead >= MIN_LOOKAHEAD or flush:
            if self.lookahead == 0:
                if self.prevAvailable:
                    self.huffman.TallyLit(self.window[self.strstart - 1] & 0xFF)
                self.prevAvailable = False
                self.huffman.FlushBlock(self.window, self.blockStart, self.strstart - self.blockStart, finish)
                self.blockStart = self.strstart
                return False
            if self.strstart >= 2 * WSIZE - MIN_LOOKAHEAD:
                self.SlideWindow()
            prevMatch = self.matchStart
            prevLen = self.matchLen
            if self.lookahead >= MIN_MATCH:
                hashHead = self.InsertString()
                if self.strategy != HuffmanOnly and hashHead != 0 and self.strstart - hashHead <= MAX_DIST and \
                        self.FindLongestMatch(hashHead):
                    if self.matchLen <= 5 and (self.strategy == Filtered or
                                               (self.matchLen == MIN_MATCH and self.strstart - self.matchStart > TooFar)):
                        self.matchLen = MIN_MATCH - 1
            if prevLen >= MIN_MATCH and self.matchLen <= prevLen:
                self.huffman.TallyDist(self.strstart - 1 - prevMatch, prevLen)
                prevLen -= 2
                while True:
                    self.strstart += 1
                    self.lookahead -= 1
                    if self.lookahead >= MIN_MATCH:
                        self.InsertString()
                    prevLen -= 1
                    if not (prevLen > 0):
                        break
                self.strstart += 1
                self.lookahead -= 1
                self.prevAvailable = False
                self.matchLen = MIN_MATCH - 1
            else:
                if self.prevAvailable:
                    self.huffman.TallyLit(self.window[self.strstart - 1] & 0xFF)
                self.prevAvailable = True
                self.strstart += 1
                self.lookahead -= 1
            if self.huffman.IsFull():
                length = self.strstart - self.blockStart
                if self.prevAvailable:
                    length -= 1
                lastBlock = finish and (self.lookahead == 0) and not self.prevAvailable
                self.huffman.FlushBlock(self.window, self.blockStart, length, lastBlock)
                self.blockStart += length
                return not lastBlock
        return True


# ---- Deflater.cs
IS_SETDICT, IS_FLUSHING, IS_FINISHING = 0x01, 0x04, 0x08
INIT_STATE, SETDICT_STATE, BUSY_STATE, FLUSHING_STATE, FINISHING_STATE, FINISHED_STATE, CLOSED_STATE = 0x00, 0x01, 0x10, 0x14, 0x1c, 0x1e, 0x7f
BEST_COMPRESSION, BEST_SPEED, DEFAULT_COMPRESSION, NO_COMPRESSION, DEFLATED = 9, 1, -1, 0, 8


class Deflater:
    def __init__(self, level=DEFAULT_COMPRESSION, noZlibHeaderOrFooter=False):     # :178
        if level == DEFAULT_COMPRESSION:
            level = 6
        elif level < NO_COMPRESSION or level > BEST_COMPRESSION:
            raise ValueError("level")
        self.level = 0
        self.state = 0
        self.totalOut = 0
        self.pending = DeflaterPending()
        self.engine = DeflaterEngine(self.pending, noZlibHeaderOrFooter)
        self.noZlibHeaderOrFooter = noZlibHeaderOrFooter
        self.SetStrategy(Default)
        self.SetLevel(level)
        self.Reset()

    def Reset(self):
        self.state = BUSY_STATE if self.noZlibHeaderOrFooter else INIT_STATE
        self.totalOut = 0
        self.pending.Reset()
        self.engine.Reset()

    @property
    def Adler(self):
        return self.engine.Adler

    @property
    def TotalIn(self):
        return self.engine.TotalIn

    @property
    def TotalOut(self):
        return self.totalOut

    def Flush(self):
        self.state |= IS_FLUSHING

    def Finish(self):
        self.state |= (IS_FLUSHING | IS_FINISHING)

    @property
    def IsFinished(self):
        return self.state == FINISHED_STATE and self.pending.IsFlushed

    @property
    def IsNeedingInput(self):
        return self.engine.NeedsInput()

    def SetInput(self, input_, offset=0, count=None):
        if count is None:
            count = len(input_)
        if (self.state & IS_FINISHING) != 0:
            raise RuntimeError("Finish() already called")
        self.engine.SetInput(input_, offset, count)

    def SetLevel(self, level):
        if level == DEFAULT_COMPRESSION:
            level = 6
        elif level < NO_COMPRESSION or level > BEST_COMPRESSION:
            raise ValueError("level")
        if self.level != level:
            self.level = level
            self.engine.SetLevel(level)

    def GetLevel(self):
        return self.level

    def SetStrategy(self, strategy):
        self.engine.strategy = strategy

    def Deflate(self, output, offset=0, length=None):      # :427
        if length is None:
            length = len(output)
        origLength = length
        if self.state == CLOSED_STATE:
            raise RuntimeError("Deflater closed")
        if self.state < BUSY_STATE:
            header = (DEFLATED + ((MAX_WBITS - 8) << 4)) << 8
            level_flags = (self.level - 1) >> 1
            if level_flags < 0 or level_flags > 3:
                level_flags = 3
            header |= level_flags << 6
            if (self.state & IS_SETDICT) != 0:
                header |= PRESET_DICT
            header += 31 - (header % 31)
            self.pending.WriteShortMSB(header)
            if (self.state & IS_SETDICT) != 0:
                chksum = self.engine.Adler
                self.engine.ResetAdler()
                self.pending.WriteShortMSB(chksum >> 16)
                self.pending.WriteShortMSB(chksum & 0xFFFF)
            self.state = BUSY_STATE | (self.state & (IS_FLUSHING | IS_FINISHING))
        while True:
            count = self.pending.Flush(output, offset, length)
            offset += count
            self.totalOut += count
            length -= count
            if length == 0 or self.state == FINISHED_STATE:
                break
            if not self.engine.Deflate((self.state & IS_FLUSHING) != 0, (self.state & IS_FINISHING) != 0):
                if self.state == BUSY_STATE:
                    return origLength - length
                elif self.state == FLUSHING_STATE:
                    if self.level != NO_COMPRESSION:
                        neededbits = 8 + ((-self.pending.BitCount) & 7)
                        while neededbits > 0:
                            self.pending.WriteBits(2, 10)
                            neededbits -= 10
                    self.state = BUSY_STATE
                elif self.state == FINISHING_STATE:
                    self.pending.AlignToByte()
                    if not self.noZlibHeaderOrFooter:
                        adler = self.engine.Adler
                        self.pending.WriteShortMSB(adler >> 16)
                        self.pending.WriteShortMSB(adler & 0xFFFF)
                    self.state = FINISHED_STATE
        return origLength - length

    def SetDictionary(self, dictionary, index=0, count=None):
        if count is None:
            count = len(dictionary)
        if self.state != INIT_STATE:
            raise RuntimeError("invalid state")
        self.state = SETDICT_STATE
        self.engine.SetDictionary(dictionary, index, count)


def deflate(data, level=6, nowrap=True, strategy=Default, flush=False, chunk=None, dictionary=None, out_chunk=4096):
    """DeflaterOutputStream-style drive: Write(chunks) [+Flush()] + Finish(), draining with Deflate(buf) like
    CS/DeflaterOutputStream.cs:242-272,100-119."""
    d = Deflater(level, nowrap)
    d.SetStrategy(strategy)
    if dictionary is not None:
        d.SetDictionary(bytes(dictionary))
    data = bytes(data)
    out = bytearray()
    buf = bytearray(out_chunk)
    pos = 0
    chunk = chunk or max(len(data), 1)
    while pos < len(data):
        d.SetInput(data, pos, min(chunk, len(data) - pos))
        pos += min(chunk, len(data) - pos)
        while not d.IsNeedingInput:
            n = d.Deflate(buf, 0, len(buf))
            if n <= 0:
                break
            out += buf[:n]
    if flush:
        d.Flush()
        while True:
            n = d.Deflate(buf, 0, len(buf))
            if n <= 0:
                break
            out += buf[:n]
    d.Finish()
    while not d.IsFinished:
        n = d.Deflate(buf, 0, len(buf))
        if n <= 0:
            raise RuntimeError("Can't deflate all input?")
        out += buf[:n]
    return bytes(out)

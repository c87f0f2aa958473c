/*
 * szl_dotnet_random.c — restatement of .NET's seeded System.Random (the subtractive generator of
 * Knuth, "Net5CompatSeedImpl" in .NET 5+, identical to .NET Framework's Random(int)).
 * TEST INFRASTRUCTURE ONLY.  The reference's tests build their inputs with it
 * (T/TestSupport/Utils.cs:79-85 GetDummyBytes -> new Random(seed).NextBytes;
 *  T/Checksum/ChecksumTests.cs:41-61: 256 MiB of Random(1) + "123456789" -> Adler32 0xD4897DA3).
 * System.Random lives in the .NET BCL, which is not part of /root/reference and not installed here;
 * the algorithm below is its published one and is pinned by reproducing that Adler-32 known answer.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

typedef struct { int32_t SeedArray[56]; int inext, inextp; } dotnet_random;

static void dr_init(dotnet_random *r, int32_t Seed) {
    const int32_t MBIG = 2147483647, MSEED = 161803398;
    int ii = 0;
    int32_t mj, mk;
    int32_t subtraction = (Seed == INT32_MIN) ? INT32_MAX : (Seed < 0 ? -Seed : Seed);
    mj = MSEED - subtraction;
    r->SeedArray[55] = mj;
    mk = 1;
    for (int i = 1; i < 55; i++) {
        if ((ii += 21) >= 55) ii -= 55;
        r->SeedArray[ii] = mk;
        mk = mj - mk;
        if (mk < 0) mk += MBIG;
        mj = r->SeedArray[ii];
    }
    for (int k = 1; k < 5; k++) {
        for (int i = 1; i < 56; i++) {
            int n = i + 30;
            if (n >= 55) n -= 55;
            r->SeedArray[i] = (int32_t)((uint32_t)r->SeedArray[i] - (uint32_t)r->SeedArray[1 + n]); /* C# unchecked int */
            if (r->SeedArray[i] < 0) r->SeedArray[i] += MBIG;
        }
    }
    r->inext = 0;
    r->inextp = 21;
}
static int32_t dr_sample(dotnet_random *r) {
    const int32_t MBIG = 2147483647;
    int locINext = r->inext, locINextp = r->inextp;
    if (++locINext >= 56) locINext = 1;
    if (++locINextp >= 56) locINextp = 1;
    int32_t retVal = r->SeedArray[locINext] - r->SeedArray[locINextp];
    if (retVal == MBIG) retVal--;
    if (retVal < 0) retVal += MBIG;
    r->SeedArray[locINext] = retVal;
    r->inext = locINext;
    r->inextp = locINextp;
    return retVal;
}
/* new Random(seed).NextBytes(buf) */
void szo_dotnet_random_bytes(int32_t seed, uint8_t *buf, size_t n) {
    dotnet_random r;
    dr_init(&r, seed);
    for (size_t i = 0; i < n; i++) buf[i] = (uint8_t)dr_sample(&r);
}

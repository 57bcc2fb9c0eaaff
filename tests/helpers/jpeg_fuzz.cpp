// jpeg_fuzz.cpp -- test helper: decodes every file named on the command line with the drop-in's JPEG decoder (grey and
// colour).  Built with -fsanitize=address,undefined by tests/test_host_io.py: corrupt input may be refused, never read or
// written out of bounds.  Prints one line per file: "<path> grey=<0|1> colour=<0|1> WxH".
#include <cstdint>
#include <cstdio>
#include <vector>

bool DecodeJpegGray(const uint8_t *data, size_t size, std::vector<uint8_t> &gray, int &width, int &height);
bool DecodeJpegBGR(const uint8_t *data, size_t size, std::vector<uint8_t> &bgr, int &width, int &height);

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) {
            return 2;
        }
        std::vector<uint8_t> buf;
        uint8_t tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) {
            buf.insert(buf.end(), tmp, tmp + n);
        }
        fclose(f);
        // an exact-size heap copy, so that a read past the end of the file is a read past the end of an allocation
        std::vector<uint8_t> exact(buf.begin(), buf.end());
        std::vector<uint8_t> out;
        int w = 0, h = 0;
        const bool g = DecodeJpegGray(exact.data(), exact.size(), out, w, h);
        const bool c = DecodeJpegBGR(exact.data(), exact.size(), out, w, h);
        printf("%s grey=%d colour=%d %dx%d\n", argv[i], (int)g, (int)c, w, h);
    }
    return 0;
}

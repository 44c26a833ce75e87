"""Command line with the reference's interface (src/main.rs):

    python -m redux_amd.cli (-c | -d) [-i <input file>] [-o <output file>] [--block-size N]

Same flags, same fixed Parameters::new(8, 30, 32) (main.rs:108), same exit codes (1 usage,
2 cannot open a file, 3 coding error) and the same summary line on stderr (main.rs:112,117).
`--block-size 0` (the default) produces / expects the reference's raw single stream -- one
coder, so one GPU lane; any other value cuts the input into independent blocks coded in
parallel and wraps them in the container of redux_amd/container.py.  All coding runs on the
GPU: there is no CPU path.
"""
import io
import sys

USAGE = "Usage: redux (-c | -d) [-i <input file>] [-o <output file>] [--block-size <bytes>]"


def parse(argv):
    opts = {"compress": None, "input": None, "output": None, "block_size": 0}
    it = iter(argv)
    for arg in it:
        if arg == "-c":
            opts["compress"] = True
        elif arg == "-d":
            opts["compress"] = False
        elif arg in ("-i", "-o", "--block-size"):
            val = next(it, None)
            if val is None:
                return None
            if arg == "-i":
                opts["input"] = val
            elif arg == "-o":
                opts["output"] = val
            else:
                try:
                    opts["block_size"] = int(val)
                except ValueError:
                    return None
                if not 0 <= opts["block_size"] <= (1 << 30):  # container.MAX_BLOCK_SIZE: a usage error, not a traceback
                    return None
        else:
            return None
    return None if opts["compress"] is None else opts


def main(argv=None):
    opts = parse(sys.argv[1:] if argv is None else argv)
    if opts is None:
        print(USAGE, file=sys.stderr)
        return 1
    try:
        data = sys.stdin.buffer.read() if opts["input"] is None else open(opts["input"], "rb").read()
    except OSError as e:
        print(f"Error while opening input file {opts['input']}: {e}", file=sys.stderr)
        return 2
    try:
        sink = sys.stdout.buffer if opts["output"] is None else open(opts["output"], "wb")
    except OSError as e:
        print(f"Error while opening output file {opts['output']}: {e}", file=sys.stderr)
        return 2

    from . import api, container
    params = api.Parameters.new(8, 30, 32)  # main.rs:108
    try:
        if opts["compress"]:
            if opts["block_size"] == 0:
                o = io.BytesIO()
                i_n, o_n = api.compress(io.BytesIO(data), o, api.AdaptiveTreeModel.new(params))
                sink.write(o.getvalue())
            else:
                blob = container.compress_bytes(data, opts["block_size"], params)
                sink.write(blob)
                i_n, o_n = len(data), len(blob)
            print("Compressed %d bytes into %d bytes, ratio: %.3f" % (i_n, o_n, i_n / o_n), file=sys.stderr)
        else:
            # A container is recognised by a well-formed HEADER, not by its magic alone: a raw reference
            # stream may begin with the same four bytes.  Once the header is consistent the input IS a
            # container: a truncated or damaged body is reported as such (exit 3), not decoded as garbage.
            is_container = container.header_is_wellformed(data)
            if is_container:
                out = container.decompress_bytes(data)
                sink.write(out)
                i_n, o_n = len(data), len(out)
            else:
                o = io.BytesIO()
                i_n, o_n = api.decompress(io.BytesIO(data), o, api.AdaptiveTreeModel.new(params))
                sink.write(o.getvalue())
            ratio = (o_n / i_n) if i_n else float("nan")
            print("Decompressed %d bytes from %d bytes, ratio: %.3f" % (o_n, i_n, ratio), file=sys.stderr)
    except api.Error as e:
        print(("Compression" if opts["compress"] else "Decompression") + f" error: {e}", file=sys.stderr)
        return 3
    finally:
        if opts["output"] is not None:
            sink.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

# rocgdb -batch -x tools/probe/endcapture_cycle.gdb --args python tools/probe/capture_two_chain.py --child two_chains=1,...
# At the SIGSEGV inside hip::Stream::EndCapture (infinite recursion): `this` of the innermost frames (rbp is callee-saved and holds
# `this` in that function), each stream's parent (+0x2a8), origin flag (+0x2a4) and list of parallel capture streams (+0x2e0 .. +0x2e8).
set pagination off
handle SIGSEGV stop print
run
python
import gdb
seen = []
for i in range(0, 14):
    try:
        gdb.execute("frame %d" % i, to_string=True)
        this = int(gdb.parse_and_eval("$rbp")) & 0xffffffffffffffff
    except gdb.error as e:
        print("frame", i, "error", e)
        break
    seen.append(this)
print("this per frame (innermost first):", " ".join(hex(t) for t in seen))
inf = gdb.selected_inferior()
import struct
def q(addr):
    return struct.unpack("<Q", bytes(inf.read_memory(addr, 8)))[0]
for t in sorted(set(seen[1:])):
    try:
        b, e = q(t + 0x2e0), q(t + 0x2e8)
        lst = [hex(q(a)) for a in range(b, e, 8)][:16]
        origin = bytes(inf.read_memory(t + 0x2a4, 1))[0]
        print("stream %s: origin %d status %d parent %s parallel %s" % (hex(t), origin, q(t + 0x290) & 0xffffffff, hex(q(t + 0x2a8)), lst))
    except gdb.error as e:
        print("stream", hex(t), "unreadable", e)
end

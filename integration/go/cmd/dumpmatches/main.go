// dumpmatches settles the parity question this repository cannot settle by
// itself (its build image has no Go): it runs the reference's matchfinder.M4
// over a file and writes the Match list as little-endian u32 triples
// (Unmatched, Length, Distance) -- the format of tests/oracle_lib.py's
// OracleM4.FindMatches and of tests/golden/m4_matches.npz.
//
// usage: dumpmatches <input> <MaxDistance> <ChainLength> <HashLen> <DistanceBitCost> [blockSize] > matches.u32le
// With blockSize the input is fed block by block without Reset (what
// matchfinder.Writer does); without it, in one call.
// NOT COMPILED in this repository's build image.
package main

import (
	"bufio"
	"encoding/binary"
	"os"
	"strconv"

	"github.com/andybalholm/brotli/matchfinder"
)

func atoi(s string) int {
	n, err := strconv.Atoi(s)
	if err != nil {
		panic(err)
	}
	return n
}

func main() {
	data, err := os.ReadFile(os.Args[1])
	if err != nil {
		panic(err)
	}
	m := &matchfinder.M4{MaxDistance: atoi(os.Args[2]), ChainLength: atoi(os.Args[3]),
		HashLen: atoi(os.Args[4]), DistanceBitCost: atoi(os.Args[5])}
	block := len(data)
	if len(os.Args) > 6 {
		block = atoi(os.Args[6])
	}
	w := bufio.NewWriter(os.Stdout)
	defer w.Flush()
	for pos := 0; pos < len(data) || pos == 0; pos += max(block, 1) {
		end := min(pos+block, len(data))
		for _, x := range m.FindMatches(nil, data[pos:end]) {
			binary.Write(w, binary.LittleEndian, [3]uint32{uint32(x.Unmatched), uint32(x.Length), uint32(x.Distance)})
		}
		if block == 0 {
			break
		}
	}
}

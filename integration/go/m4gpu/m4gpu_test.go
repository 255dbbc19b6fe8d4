package m4gpu

// The reference's own M4 tests (brotli_test.go:626-649, 730-744) with the GPU
// match finder plugged in, plus the check the reference cannot make: the Match
// list equals matchfinder.M4's.  NOT COMPILED in this repository's build image.

import (
	"bytes"
	"io"
	"os"
	"reflect"
	"testing"

	"github.com/andybalholm/brotli"
	"github.com/andybalholm/brotli/matchfinder"
)

const opticks = "../../testdata/Isaac.Newton-Opticks.txt"

func roundTrip(t *testing.T, m matchfinder.MatchFinder, blockSize int) {
	data, err := os.ReadFile(opticks)
	if err != nil {
		t.Fatal(err)
	}
	b := new(bytes.Buffer)
	w := &matchfinder.Writer{Dest: b, MatchFinder: m, Encoder: &brotli.Encoder{}, BlockSize: blockSize}
	w.Write(data)
	w.Close()
	got, err := io.ReadAll(brotli.NewReader(bytes.NewReader(b.Bytes())))
	if err != nil {
		t.Fatal(err)
	}
	if !bytes.Equal(got, data) {
		t.Fatal("decompressed output doesn't match")
	}
}

func TestEncodeM4(t *testing.T)         { roundTrip(t, &M4{MaxDistance: 1 << 18, DistanceBitCost: 66}, 1<<16) }
func TestEncodeM4Chain1(t *testing.T)   { roundTrip(t, &M4{MaxDistance: 1 << 18, ChainLength: 1, DistanceBitCost: 66}, 1<<16) }
func TestEncodeM4Chain256(t *testing.T) { roundTrip(t, &M4{MaxDistance: 1 << 18, ChainLength: 256, DistanceBitCost: 66}, 1<<16) }

func TestSameMatchesAsReference(t *testing.T) {
	data, err := os.ReadFile(opticks)
	if err != nil {
		t.Fatal(err)
	}
	for _, chain := range []int{0, 1, 64} {
		ref := &matchfinder.M4{MaxDistance: 1 << 18, ChainLength: chain, DistanceBitCost: 66}
		gpu := &M4{MaxDistance: 1 << 18, ChainLength: chain, DistanceBitCost: 66}
		for pos := 0; pos < len(data); pos += 1 << 16 {
			blk := data[pos:min(pos+1<<16, len(data))]
			a, b := ref.FindMatches(nil, blk), gpu.FindMatches(nil, blk)
			if !reflect.DeepEqual(a, b) {
				t.Fatalf("ChainLength %d, block at %d: Match lists differ", chain, pos)
			}
		}
	}
}

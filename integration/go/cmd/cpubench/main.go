// cpubench is the CPU baseline SURVEY.md section 8(d) asks for: the reference's
// matchfinder.M4 over the benchmark's chunks, one M4 per worker goroutine,
// Reset per chunk (AutoReset semantics, textencoder.go:46-53), FindMatches
// only.  It prints one JSON line with input GB/s for `workers` goroutines and
// for one.  bench.py reports the C restatement instead when Go is unavailable
// (the build image of this repository has none, so this file is NOT COMPILED
// there).
//
// usage: cpubench <file of concatenated chunks> <chunkBytes> <MaxDistance> <ChainLength> <HashLen> <DistanceBitCost> [workers]
package main

import (
	"fmt"
	"os"
	"runtime"
	"strconv"
	"sync"
	"time"

	"github.com/andybalholm/brotli/matchfinder"
)

func atoi(s string) int {
	n, err := strconv.Atoi(s)
	if err != nil {
		panic(err)
	}
	return n
}

func run(data []byte, chunk, workers int, mk func() *matchfinder.M4) (seconds float64, matches int64) {
	nchunks := (len(data) + chunk - 1) / chunk
	counts := make([]int64, workers)
	var wg sync.WaitGroup
	t0 := time.Now()
	for w := 0; w < workers; w++ {
		wg.Add(1)
		go func(w int) {
			defer wg.Done()
			m := mk()
			var dst []matchfinder.Match
			for c := w; c < nchunks; c += workers {
				m.Reset()
				dst = m.FindMatches(dst[:0], data[c*chunk:min((c+1)*chunk, len(data))])
				counts[w] += int64(len(dst))
			}
		}(w)
	}
	wg.Wait()
	seconds = time.Since(t0).Seconds()
	for _, c := range counts {
		matches += c
	}
	return
}

func main() {
	data, err := os.ReadFile(os.Args[1])
	if err != nil {
		panic(err)
	}
	chunk := atoi(os.Args[2])
	mk := func() *matchfinder.M4 {
		return &matchfinder.M4{MaxDistance: atoi(os.Args[3]), ChainLength: atoi(os.Args[4]),
			HashLen: atoi(os.Args[5]), DistanceBitCost: atoi(os.Args[6])}
	}
	workers := runtime.NumCPU()
	if len(os.Args) > 7 {
		workers = atoi(os.Args[7])
	}
	sN, mN := run(data, chunk, workers, mk)
	s1, _ := run(data[:min(len(data), 64*chunk)], chunk, 1, mk)
	fmt.Printf("{\"kind\": \"reference\", \"GOMAXPROCS\": %d, \"workers\": %d, \"input_bytes\": %d, \"matches\": %d, "+
		"\"value\": %.4f, \"value_1worker\": %.4f, \"unit\": \"GB/s\"}\n",
		runtime.GOMAXPROCS(0), workers, len(data), mN, float64(len(data))/sN/1e9,
		float64(min(len(data), 64*chunk))/s1/1e9)
}
